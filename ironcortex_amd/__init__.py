"""ironcortex_amd — MI355X-native hot path of IronCortex behind the reference's Python API
(exports mirror reference ironcortex/__init__.py)."""
from .config import CortexConfig as CortexConfig
from .model import CortexReasoner as CortexReasoner
from .training import LossWeights as LossWeights, train_step as train_step
from .generation import (generate as generate, DiffusionConfig as DiffusionConfig,
                         diffusion_generate as diffusion_generate)
from .energy import (EnergyVerifierHead as EnergyVerifierHead, energy_descent_step as energy_descent_step,
                     ff_energy_loss as ff_energy_loss, ff_step as ff_step, Thinker as Thinker)
from .evaluation import evaluate_perplexity as evaluate_perplexity
from .wiring import hex_axial_coords as hex_axial_coords, hex_neighbors as hex_neighbors
from .sdr import SparseTokenEncoder as SparseTokenEncoder
from .checkpoint import save_checkpoint as save_checkpoint, load_checkpoint as load_checkpoint
from .region import RWKVRegionCell as RWKVRegionCell
from .gate import Gate as Gate, Router as Router
from .utils import KWTA as KWTA, RMSNorm as RMSNorm, nms_topk as nms_topk
