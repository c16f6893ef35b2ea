"""Device-resident voxelize -> CSG -> JFA pipeline on one GPU.

torch is plumbing only (device memory + stream); every computation goes through the C ABI
(capi.Context -> libvphip.so).  Mirrors the flow of the reference CLI
(/root/reference/apps/cli/main.cpp:62-218): one frame shared by all meshes, one grid per mesh,
CSG accumulated into grid 0, JFA on grid 0 with an -inf pre-fill.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import capi
from .capi import ALGO_NAIVE, ALGO_TILED, Frame  # noqa: F401


class Engine:
    def __init__(self, device: int = 0):
        if not torch.cuda.is_available():
            raise RuntimeError("Engine needs a GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.ctx = capi.Context(device)
        self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream, external=True)
        self._work = None

    # -- buffers ---------------------------------------------------------------------------
    def to_device(self, arr, dtype):
        t = torch.from_numpy(np.array(arr, dtype=dtype, order="C", copy=True).view(np.int32 if dtype == np.uint32 else dtype))
        return t.to(self.device)

    def mesh_to_device(self, xyz, tri):
        return self.to_device(xyz, np.float32), self.to_device(tri, np.uint32)

    def new_grid(self, frame: Frame):
        return torch.empty(frame.words, dtype=torch.int32, device=self.device)

    def _workspace(self, nbytes: int):
        if self._work is None or self._work.numel() < nbytes:
            self._work = None
            self._work = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._work

    # -- stages ----------------------------------------------------------------------------
    def voxelize(self, frame: Frame, d_xyz, d_tri, out=None, algo=ALGO_TILED, accumulate=False):
        if out is None:
            out = self.new_grid(frame)
            if accumulate:
                out.zero_()
        self.ctx.voxelize(frame, out.data_ptr(), d_xyz.data_ptr(), d_xyz.shape[0], d_tri.data_ptr(), d_tri.shape[0],
                          algo, accumulate)
        return out

    def csg(self, a, b, op: int):
        self.ctx.csg(a.data_ptr(), b.data_ptr(), a.numel(), op)
        return a

    def jfa(self, frame: Frame, words, out=None, fill=-math.inf, algo=ALGO_TILED):
        if out is None:
            out = torch.empty(frame.voxels, dtype=torch.float32, device=self.device)
        nb = self.ctx.jfa_workspace_bytes(frame)
        work = self._workspace(nb)
        self.ctx.jfa(frame, words.data_ptr(), fill, out.data_ptr(), work.data_ptr(), nb, algo)
        return out

    def surface(self, frame: Frame, words, out=None):
        if out is None:
            out = self.new_grid(frame)
        self.ctx.surface(frame, words.data_ptr(), None, None, out.data_ptr())
        return out

    def sync(self):
        self.ctx.sync()

    @staticmethod
    def words_to_numpy(t):
        return t.detach().cpu().numpy().view(np.uint32)
