import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import pkgload; pkg = pkgload.load()
import torch
for name, n in (("C2", 1024), ("C4", 8192)):
    p = pkg.petlion(pkg.LCO)
    cfg = getattr(pkg.configs, name.lower())(p, n)
    Th = np.ascontiguousarray(cfg["theta"])
    Thd = torch.from_numpy(Th).cuda()
    for _ in range(3):
        e = pkg.simulate_ensemble(p, Thd, cfg["protocol"], SOC=cfg["SOC"], device=True, max_points=cfg["max_points"]); torch.cuda.synchronize()
    kms = e.kernel_ms
    ref = pkg.simulate_ensemble(p, Th, cfg["protocol"], SOC=cfg["SOC"], max_points=cfg["max_points"])
    ts = []
    for _ in range(7):
        t1 = time.perf_counter(); h = pkg.simulate_ensemble(p, Th, cfg["protocol"], SOC=cfg["SOC"], max_points=cfg["max_points"]); ts.append(time.perf_counter() - t1)
    ok = np.array_equal(np.asarray(h.Y), e.Y.cpu().numpy()) and np.array_equal(h.n_pts, e.n_pts.cpu().numpy() if hasattr(e.n_pts, 'cpu') else e.n_pts)
    i = 5; k = int(h.n_pts[i])
    okt = np.array_equal(np.asarray(h.t)[i, :k], (e.t.cpu().numpy() if hasattr(e.t, 'cpu') else e.t)[i, :k])
    print("%s: kernel %.3f ms, blocking host call median %.3f ms (min %.3f) -> %.2f of the kernel rate; results equal: %s %s" % (name, kms, 1e3 * np.median(ts), 1e3 * min(ts), kms / (1e3 * np.median(ts)), ok, okt))
