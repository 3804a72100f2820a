"""Development aid (CPU): the Newton-Schulz smoother's iteration counts at the headline shape with the kernel's rules
(kf_smoother_b4.hip: third-order step above 1e-7, break after the update when rnorm <= 2e-5), with the previous inverse
as warm start (order 0), linearly (1) and quadratically (2) extrapolated warm starts.  Products per step beyond the six."""
import sys
import numpy as np
sys.path.insert(0, ".")
import workloads as W
from oracle import kalman_numpy as kn


def run(w, b, order):
    T, R, Q = w["T"][b], w["R"][b], w["Q"][b]
    m = T.shape[0]
    flt = kn.kalman_filter(w["y"], *[w[k][b:b + 1] for k in ("a0", "P0", "T", "Z", "R", "H", "Q")])
    fP = flt["filtered_covs"][0]
    n = fP.shape[0]
    RQR = R @ Q @ R.T
    I = np.eye(m)
    hist = []          # final inverses, most recent first
    prods = []
    its = []
    for t in range(n - 2, -1, -1):
        A = T @ fP[t] @ T.T + RQR
        X = None
        np_ = 0
        if hist:
            if order >= 2 and len(hist) >= 3:
                X = 3 * hist[0] - 3 * hist[1] + hist[2]
            elif order >= 1 and len(hist) >= 2:
                X = 2 * hist[0] - hist[1]
            else:
                X = hist[0]
            Rm = I - A @ X
            rn = np.abs(Rm).max()
            if not (rn < 1e-2) and not (np.abs(Rm).sum(1).max() < 0.7):
                X = None
        if X is None:
            hist = []
            X = I / np.abs(A).sum(0).max()
            Rm = I - A @ X
            rn = np.abs(Rm).max()
            np_ += 1
        k = 0
        while not (rn <= 1e-13):
            third = not (rn <= 1e-7)
            if third:
                Rm = Rm + Rm @ Rm
                np_ += 1
            X = X + X @ Rm
            np_ += 1
            k += 1
            if rn <= 2e-5:
                break
            rp = rn
            Rm = I - A @ X
            rn = np.abs(Rm).max()
            np_ += 1
            if rn > 0.5 * rp and rn <= 1e-7:
                break
            if k > 64:
                break
        hist = [X] + hist[:2]
        prods.append(np_)
        its.append(k)
    return np.array(its), np.array(prods)


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "headline"
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    w = W.WORKLOADS[name](B=nb)
    for b in range(nb):
        line = f"{name} draw {b:3d}:"
        for order in (0, 1, 2):
            its, pr = run(w, b, order)
            line += f"  order {order}: it/step {its.mean():.2f} prod/step {pr.mean():.2f} (last100 {pr[:100].mean():.2f})"
        print(line, flush=True)
