"""fp32 quadrotor (c4's shape): the fp32 tile kernel against the fp64 tile kernel's float instantiation over batch size and max_iter
(cost_update_thre 1e-3) — the data behind ModelOpsTile32::useTile64Float.     python scripts/c4_dispatch_sweep.py
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import c4_iteration_profile as P  # noqa: E402

for B in (64, 256, 1024, 2048, 4096, 8192, 16384):
    for mi in (2, 4, 8):
        row = []
        for kernel in ("tile32", "tile64"):
            name, (opt, bw, fw), mean_it, _ = P.run(kernel, B, mi, True, 1e-3)
            row.append((name, opt, mean_it))
        print(f"B {B:6d} max_iter {mi}: " + "   ".join(f"{n} {o:.3f} ms ({m / o * 1e3:.0f} it/s)" for n, o, m in row)
              + f"   ratio {row[0][1] / row[1][1]:.2f}", flush=True)
