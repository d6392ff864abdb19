#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the reference tree.

Run ONCE in the build container (where /root/reference exists); the resulting
.npz files are committed and are what travels to the GPU box.  Nothing in
tests/, smoke() or bench.py reads /root/reference at run time.

What is generated, and from what:

* freq_reference_torch.npz   -- outputs of the reference's pure-torch
  ``encoding.FreqEncoder`` (encoding.py:5-43; same layout as the CUDA kernel,
  constructed as the commented line encoding.py:55 says) for the two encoders
  dnerf uses (xyz: D=3, deg 10; time: D=1, deg 6), plus autograd input grads.
* trunc_exp_reference_torch.npz -- ``activation.trunc_exp`` forward/backward
  (activation.py:5-17), imported and run on CPU.
* sh_reference_closed_form.npz -- the reference's closed-form real-SH table
  (shencoder/src/shencoder.cu:49-121 and the dx/dy/dz tables :130-353) READ AS
  TEXT: each ``outputs[k] = <polynomial>;`` right-hand side is parsed and
  evaluated with numpy float64 on seeded unit (and non-unit) vectors.  No CUDA
  code is compiled or executed; the fixture holds inputs and expected outputs.

The raymarching and grid kernels have no runnable form here (CUDA only, see
DESIGN.md), so they have no reference-generated fixture: "parity unpinned".
Their oracle fixtures (oracle_*.npz) are produced by gen_oracle_fixtures.py.
"""
import os
import re
import sys

sys.dont_write_bytecode = True  # never drop .pyc files into the read-only reference tree

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def gen_freq_and_activation():
    sys.path.insert(0, REF)
    import encoding as ref_encoding  # pure torch
    import activation as ref_activation  # pure torch (custom_fwd/custom_bwd decorators only)

    g = torch.Generator().manual_seed(1234)
    out = {}
    for name, D, deg, B in (("xyz", 3, 10, 257), ("time", 1, 6, 33)):
        enc = ref_encoding.FreqEncoder(input_dim=D, max_freq_log2=deg - 1, N_freqs=deg, log_sampling=True)
        x = (torch.rand(B, D, generator=g, dtype=torch.float32) * 2 - 1).requires_grad_(True)
        y = enc(x)
        gy = torch.randn(y.shape, generator=g, dtype=torch.float32)
        (gx,) = torch.autograd.grad(y, x, gy)
        out[f"{name}_inputs"] = x.detach().numpy()
        out[f"{name}_outputs"] = y.detach().numpy()
        out[f"{name}_grad_outputs"] = gy.numpy()
        out[f"{name}_grad_inputs"] = gx.numpy()
        out[f"{name}_degree"] = np.int32(deg)
    np.savez(os.path.join(OUT, "freq_reference_torch.npz"), **out)

    x = torch.linspace(-20, 20, 161, dtype=torch.float32).requires_grad_(True)
    y = ref_activation.trunc_exp(x)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float32)
    (gx,) = torch.autograd.grad(y, x, gy)
    np.savez(os.path.join(OUT, "trunc_exp_reference_torch.npz"), x=x.detach().numpy(), y=y.detach().numpy(),
             grad_y=gy.numpy(), grad_x=gx.numpy())


_LINE = re.compile(r"^\s*(outputs|dx|dy|dz)\[(\d+)\]\s*=\s*(.*?);")


def gen_sh():
    text = open(os.path.join(REF, "shencoder/src/shencoder.cu")).read().splitlines()
    exprs = {"outputs": {}, "dx": {}, "dy": {}, "dz": {}}
    for line in text:
        m = _LINE.match(line)
        if not m:
            continue
        rhs = m.group(3)
        rhs = re.sub(r"(\d+\.\d*(?:[eE][+-]?\d+)?)f", r"\1", rhs)  # 1.0f -> 1.0
        rhs = re.sub(r"pow\(\s*z\s*,\s*3\s*\)", "(z**3)", rhs)
        exprs[m.group(1)][int(m.group(2))] = rhs
    assert all(len(exprs[k]) == 64 for k in exprs), {k: len(v) for k, v in exprs.items()}

    rng = np.random.default_rng(4321)
    v = rng.standard_normal((512, 3))
    v[:384] /= np.linalg.norm(v[:384], axis=1, keepdims=True)  # unit vectors ...
    v[384:] *= 0.7  # ... and some non-unit ones (the kernels are plain polynomials)
    v[0] = (0, 0, 1)
    v[1] = (1, 0, 0)
    v[2] = (0, -1, 0)
    v = v.astype(np.float32).astype(np.float64)
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    env = dict(x=x, y=y, z=z, xy=x * y, xz=x * z, yz=y * z, x2=x * x, y2=y * y, z2=z * z, xyz=x * y * z)
    env.update(x4=env["x2"] ** 2, y4=env["y2"] ** 2, z4=env["z2"] ** 2)
    env.update(x6=env["x4"] * env["x2"], y6=env["y4"] * env["y2"], z6=env["z4"] * env["z2"])
    res = {}
    for k in exprs:
        arr = np.zeros((v.shape[0], 64))
        for i in range(64):
            arr[:, i] = eval(exprs[k][i], {"__builtins__": {}}, env) + np.zeros_like(x)
        res[k] = arr
    np.savez(os.path.join(OUT, "sh_reference_closed_form.npz"), inputs=v.astype(np.float32), outputs=res["outputs"],
             dx=res["dx"], dy=res["dy"], dz=res["dz"])


if __name__ == "__main__":
    gen_freq_and_activation()
    gen_sh()
    print("wrote", sorted(f for f in os.listdir(OUT) if f.endswith(".npz")))
