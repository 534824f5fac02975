// HIP graph capture / replay of a launch sequence of this library (ic_graph_* in the header).
// The device-steered reasoning loops (ic_loop_steer) take no decision on the host, so the ~1300 launches of a
// train_step's loops are the same launch sequence every step: captured once, replayed with one call.
#include "ic_common.h"

extern "C" int ic_graph_begin(ic_stream_t stream) {
  IC_REQUIRE(stream != nullptr, "ic_graph_begin: capture needs a stream of its own (not the default stream)");
  const hipError_t e = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeRelaxed);
  if (e != hipSuccess) {
    ic_set_error("ic_graph_begin: hipStreamBeginCapture: %s", hipGetErrorString(e));
    return IC_ELAUNCH;
  }
  return IC_OK;
}

extern "C" int ic_graph_end(ic_stream_t stream, void** exec_out, int32_t* n_nodes) {
  IC_REQUIRE(stream != nullptr && exec_out != nullptr, "ic_graph_end: NULL argument");
  *exec_out = nullptr;
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture((hipStream_t)stream, &graph);
  if (e != hipSuccess || graph == nullptr) {
    ic_set_error("ic_graph_end: hipStreamEndCapture: %s", hipGetErrorString(e));
    return IC_ELAUNCH;
  }
  if (n_nodes) {
    size_t n = 0;
    (void)hipGraphGetNodes(graph, nullptr, &n);
    *n_nodes = (int32_t)n;
  }
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess || exec == nullptr) {
    ic_set_error("ic_graph_end: hipGraphInstantiate: %s", hipGetErrorString(e));
    return IC_ELAUNCH;
  }
  *exec_out = (void*)exec;
  return IC_OK;
}

extern "C" int ic_graph_launch(void* exec, ic_stream_t stream) {
  IC_REQUIRE(exec != nullptr, "ic_graph_launch: NULL graph");
  const hipError_t e = hipGraphLaunch((hipGraphExec_t)exec, (hipStream_t)stream);
  if (e != hipSuccess) {
    ic_set_error("ic_graph_launch: hipGraphLaunch: %s", hipGetErrorString(e));
    return IC_ELAUNCH;
  }
  return IC_OK;
}

extern "C" int ic_graph_destroy(void* exec) {
  if (exec) (void)hipGraphExecDestroy((hipGraphExec_t)exec);
  return IC_OK;
}
