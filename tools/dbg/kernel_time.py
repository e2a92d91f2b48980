import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pkgload, torch
pkg = pkgload.load()
p = pkg.petlion(pkg.LCO)
Th = torch.from_numpy(pkg.theta_matrix(p, 1024)).cuda()
ms = []
for k in range(30):
    ens = pkg.simulate_ensemble(p, Th, [{"I": -1.0}], SOC=1.0, device=True, max_points=256)
    torch.cuda.synchronize()
    if k >= 10: ms.append(ens.kernel_ms)
print("kernel ms: mean %.4f min %.4f ; flags %s" % (np.mean(ms), np.min(ms), np.unique(ens.run_info["flag"])))
