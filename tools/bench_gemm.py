#!/usr/bin/env python3
"""Timing of the training path's three GEMM shapes (forward, dX, dW) at the size of one training step
(655,360 rows x 512 x 512), both precisions.  Diagnostic; prints one line per shape."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from diner_amd import training as T  # noqa: E402


def main(M=655360, reps=3):
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    X = torch.randn((M, 512), device=dev, generator=g)
    W = torch.randn((512, 512), device=dev, generator=g) * 0.06
    b = torch.zeros(512, device=dev)
    dY = torch.randn((M, 512), device=dev, generator=g) * 1e-3
    out = torch.empty((M, 512), device=dev)
    dw = torch.zeros((512, 512), device=dev)
    flop = 2.0 * M * 512 * 512
    for prec in (0, 1):
        a = T.amax_of(dY, prec)
        cases = {
            "fwd": lambda: T.linear_fwd(X, W, b, out, relu_in=True, prec=prec),
            "fwd+acc": lambda: T.linear_fwd(X, W, b, out, relu_in=True, addend=out, prec=prec),
            "dX": lambda: T.linear_bwd_x(dY, W, X, out, prec=prec, amax=a),
            "dW": lambda: T.linear_bwd_w(dY, X, dw, None, relu_x=True, prec=prec, amax=a),
        }
        if prec:
            T.USE_CORE = False
            wp, wtp = T.split_panel(W, False, prec), T.split_panel(W, True, prec)
            T.USE_CORE = True
            wc, wtc = T.split_panel(W, False, prec), T.split_panel(W, True, prec)
            cases["fwd core"] = lambda: T.linear_fwd(X, W, b, out, relu_in=True, prec=prec, panel=wc)
            cases["fwd+add core"] = lambda: T.linear_fwd(X, W, b, out, relu_in=True, addend=out, prec=prec, panel=wc)
            cases["dX core"] = lambda: T.linear_bwd_x(dY, W, X, out, prec=prec, amax=a, panel=wtc)
            cases["fwd panel"] = lambda: T.linear_fwd(X, W, b, out, relu_in=True, prec=prec, panel=wp)
            cases["fwd+add panel"] = lambda: T.linear_fwd(X, W, b, out, relu_in=True, addend=out, prec=prec, panel=wp)
            cases["dX panel"] = lambda: T.linear_bwd_x(dY, W, X, out, prec=prec, amax=a, panel=wtp)
        for name, fn in cases.items():
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            print(f"prec={'f16x3' if prec else 'fp32'} {name:14s} {ms:7.3f} ms  {flop / ms / 1e9:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
