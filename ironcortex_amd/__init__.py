"""ironcortex_amd — MI355X-native hot path of IronCortex behind the reference's Python API."""
from .config import CortexConfig as CortexConfig
from .wiring import hex_axial_coords as hex_axial_coords, hex_neighbors as hex_neighbors
from .sdr import SparseTokenEncoder as SparseTokenEncoder
