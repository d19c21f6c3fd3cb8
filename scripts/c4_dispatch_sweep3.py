"""As c4_dispatch_sweep.py on FULL chips (8192 / 16384 instances) over cost_update_thre: where the float instantiation's line search
of passes (first two step sizes; the later ones; the taken one) loses to the fp32 tile kernel's every-step-size-at-once — the
threshold side of ModelOpsTile32::useTile64Float."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import c4_iteration_profile as P  # noqa: E402

for thre in (1e-3, 3e-4, 1e-4, 1e-5):
    for B in (8192, 16384):
        for mi in (1, 2, 4, 8):
            row = []
            for kernel in ("tile32", "tile64"):
                name, (opt, bw, fw), mean_it, _ = P.run(kernel, B, mi, True, thre)
                row.append((name, opt, mean_it))
            print(f"B {B:6d} max_iter {mi} thre {thre:g}: " + "   ".join(f"{n} {o:.3f} ms ({m / o * 1e3:.0f} it/s)" for n, o, m in row)
                  + f"   ratio {row[0][1] / row[1][1]:.2f}", flush=True)
