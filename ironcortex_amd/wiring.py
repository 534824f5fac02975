"""Hex-grid wiring: axial spiral coordinates and 6-direction neighbour lists.

Static graph, built once on the host (reference wiring.py:14-67).  Cells are numbered ring by
ring; each ring starts on the east side and is walked clockwise.
"""
from typing import Dict, List, Tuple

import torch

# one ring = six straight sides; side i is walked with step _SIDE_STEPS[i]
_SIDE_STEPS = ((0, -1), (-1, 0), (-1, 1), (0, 1), (1, 0), (1, -1))
# neighbour order: E, NE, NW, W, SW, SE
_NEIGHBOR_STEPS = ((1, 0), (1, -1), (0, -1), (-1, 0), (-1, 1), (0, 1))


def _spiral_cells(R: int) -> List[Tuple[int, int]]:
    cells: List[Tuple[int, int]] = []
    ring = 0
    while len(cells) < R:
        if ring == 0:
            cells.append((0, 0))
        else:
            q, r = ring, 0
            for dq, dr in _SIDE_STEPS:
                for _ in range(ring):
                    cells.append((q, r))
                    q, r = q + dq, r + dr
        ring += 1
    return cells[:R]


def hex_axial_coords(R: int) -> torch.Tensor:
    """[R,2] float tensor of axial (q, r) coordinates."""
    if R <= 0:
        return torch.zeros((0, 2), dtype=torch.float32)
    return torch.tensor(_spiral_cells(R), dtype=torch.float32)


def hex_neighbors(R: int) -> List[List[int]]:
    """Adjacency lists; boundary cells simply have fewer neighbours."""
    cells = _spiral_cells(max(R, 0))
    where: Dict[Tuple[int, int], int] = {c: i for i, c in enumerate(cells)}
    return [[where[(q + dq, r + dr)] for dq, dr in _NEIGHBOR_STEPS if (q + dq, r + dr) in where]
            for q, r in cells]
