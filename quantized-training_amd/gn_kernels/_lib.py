"""Operator registry: the same `gn_kernels` torch.library namespace the reference creates
(reference kernels/gn_kernels/_lib.py:3-4), so `torch.ops.gn_kernels.*` names are unchanged."""
import torch

lib = torch.library.Library("gn_kernels", "DEF")
lib_ops = torch.ops.gn_kernels
