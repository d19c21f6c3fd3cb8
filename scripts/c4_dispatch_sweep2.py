"""As c4_dispatch_sweep.py at the reference's default cost_update_thre (1e-7: below what a float cost resolves — the solves keep
iterating in rounding noise and nearly every line search back-tracks through the list)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import c4_iteration_profile as P  # noqa: E402

for B in (32, 256, 1024, 4096, 8192):
    for mi in (2, 8):
        row = []
        for kernel in ("tile32", "tile64"):
            name, (opt, bw, fw), mean_it, _ = P.run(kernel, B, mi, True, 1e-7)
            row.append((name, opt, mean_it))
        print(f"B {B:6d} max_iter {mi} thre 1e-7: " + "   ".join(f"{n} {o:.3f} ms ({m / o * 1e3:.0f} it/s)" for n, o, m in row)
              + f"   ratio {row[0][1] / row[1][1]:.2f}", flush=True)
