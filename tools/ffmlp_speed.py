"""Speed of the fused MLP operator vs the same network as bias-free nn.Linear layers under autocast -- the comparison
of the reference's testing/test_ffmlp.py:100-234 (B = 2^21, 16 -> 64 x2 -> 16), plus the dnerf-sized 128-wide case.
Prints one JSON line per configuration.  Usage: python tools/ffmlp_speed.py [--iters 20]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seald-nerf_amd"))
import ffmlp  # noqa: E402


def _time(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=2 ** 21)
    ap.add_argument("--only", type=int, default=-1, help="index of the single configuration to run")
    args = ap.parse_args()
    B = args.batch
    configs = [(16, 16, 64, 2), (32, 16, 64, 4), (32, 16, 128, 8), (32, 16, 256, 4), (16, 16, 16, 2), (32, 16, 32, 3)]
    if args.only >= 0:
        configs = configs[args.only:args.only + 1]
    for in_dim, out_dim, hidden, L in configs:
        net = ffmlp.FFMLP(in_dim, out_dim, hidden, L).cuda()
        lin = torch.nn.Sequential()
        dims = [in_dim] + [hidden] * L + [out_dim]
        for i in range(L + 1):
            lin.append(torch.nn.Linear(dims[i], dims[i + 1], bias=False))
            if i != L:
                lin.append(torch.nn.ReLU())
        lin = lin.cuda()
        x = torch.rand(B, in_dim, device="cuda") * 10
        xh = x.half()
        wh = net.weights.detach().half()

        def infer_ff():
            return ffmlp.ffmlp_forward(xh, wh, in_dim, 16, hidden, L, 0, 6, True, False)

        def infer_lin():
            with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
                return lin(x)

        def train_ff():
            net.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.float16):
                y = net(x)
            y.sum().backward()

        def train_lin():
            lin.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.float16):
                y = lin(x)
            y.sum().backward()

        net.train()
        flops = 2.0 * B * hidden * (in_dim + hidden * (L - 1) + 16)
        t_if, t_il = _time(infer_ff, args.iters), _time(infer_lin, args.iters)
        t_tf, t_tl = _time(train_ff, args.iters), _time(train_lin, args.iters)
        print(json.dumps({"config": f"{in_dim}->{hidden}x{L}->{out_dim}", "batch": B,
                          "ffmlp_inference_ms": round(t_if, 4), "linear_inference_ms": round(t_il, 4),
                          "ffmlp_inference_tflops": round(flops / t_if / 1e9, 1),
                          "ffmlp_train_step_ms": round(t_tf, 4), "linear_train_step_ms": round(t_tl, 4)}), flush=True)


if __name__ == "__main__":
    main()
