"""GGXColocatedRenderer with the reference's surface (models/renderer_ggx.py:61-146), computed by
the HIP kernel `iron_ggx_colocated` (csrc/pointwise.hip, csrc/ggx_core.h).

The two Mitsuba rough-transmittance tables the reference reads from models/ggx/*.txt ship here as
iron_amd/data/mts_rtrans_tables.npz (same 5000 + 50 fp32 values).
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "mts_rtrans_tables.npz")


def load_mts_tables():
    """(MTS_TRANS [5000], MTS_DIFF_TRANS [50]) as CPU fp32 tensors (renderer_ggx.py:65-74)."""
    z = np.load(_DATA, allow_pickle=False)
    return torch.from_numpy(z["ext_rtrans"].astype(np.float32)), torch.from_numpy(z["int_diff_rtrans"].astype(np.float32))


class GGXColocatedRenderer(nn.Module):
    def __init__(self, use_cuda=False):
        super().__init__()
        a, b = load_mts_tables()
        self.MTS_TRANS, self.MTS_DIFF_TRANS = a, b
        self.num_theta_samples = 100
        self.num_alpha_samples = 50
        if use_cuda:
            self.MTS_TRANS = self.MTS_TRANS.cuda()
            self.MTS_DIFF_TRANS = self.MTS_DIFF_TRANS.cuda()

    def _tables_on(self, device):
        if self.MTS_TRANS.device != device:
            self.MTS_TRANS = self.MTS_TRANS.to(device)
            self.MTS_DIFF_TRANS = self.MTS_DIFF_TRANS.to(device)
        return self.MTS_TRANS, self.MTS_DIFF_TRANS

    def forward(self, light, distance, normal, viewdir, params={}):
        """light: scalar; distance [...,1]; normal, viewdir [...,3]; params: diffuse_albedo [...,3],
        specular_albedo [...,3], specular_roughness [...,1] -> diffuse_rgb, specular_rgb, rgb [...,3]."""
        nrm = _lib.require_cuda_f32(normal.detach(), "normal")
        sh = list(nrm.shape[:-1])
        nrm = nrm.reshape(-1, 3)
        n = nrm.shape[0]
        dev = nrm.device
        dist = _lib.require_cuda_f32(distance.detach(), "distance").reshape(-1)
        vd = _lib.require_cuda_f32(viewdir.detach(), "viewdir").reshape(-1, 3)
        kd = _lib.require_cuda_f32(params["diffuse_albedo"].detach(), "diffuse_albedo").reshape(-1, 3)
        ks = _lib.require_cuda_f32(params["specular_albedo"].detach().expand(sh + [3]), "specular_albedo").reshape(-1, 3)
        al = _lib.require_cuda_f32(params["specular_roughness"].detach(), "specular_roughness").reshape(-1)
        t1, t2 = self._tables_on(dev)
        out = [torch.empty((n, 3), dtype=torch.float32, device=dev) for _ in range(3)]
        with torch.cuda.device(dev):
            _lib.check(_lib.load().iron_ggx_colocated(float(light), dist.data_ptr(), nrm.data_ptr(), vd.data_ptr(),
                                                      kd.data_ptr(), ks.data_ptr(), al.data_ptr(), t1.data_ptr(),
                                                      t2.data_ptr(), n, out[0].data_ptr(), out[1].data_ptr(),
                                                      out[2].data_ptr(), _lib.stream_ptr(dev)))
        return {"diffuse_rgb": out[0].reshape(sh + [3]), "specular_rgb": out[1].reshape(sh + [3]),
                "rgb": out[2].reshape(sh + [3])}
