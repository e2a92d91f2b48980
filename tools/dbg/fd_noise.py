"""DESIGN.md section 5, link 1: the device and oracle residuals of the finite-difference estimate of YP_alg differ by ~5e-13; exact (long double) solves of the SAME
matrix with either residual differ by ~1e-6 in YP_Phi_e.  usage: python tools/dbg/fd_noise.py  (wave-emulator build, CPU)"""
import sys, numpy as np
import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'wave_emu'))
import pkgload, parity, build_emu
from oracle import oracle as O
pkg = pkgload.load()
p = pkg.petlion(pkg.LCO, _lib_path=build_emu.build())
cell = 164 * 256
th = np.ascontiguousarray(pkg.configs.sweep_theta(p, np.array([cell]), 4)[0])
N = p.N.tot; Nd = p.N.diff
Y0 = O.initial_guess("lco_iso", th, 1.0); Y0[-1] = -1.0
rc, Yo, YPo, ito = O.init_consistent("lco_iso", th, Y0, 0, -1.0)
Y, YP = Y0.copy(), np.zeros(N); st, it = np.zeros(1, np.int32), np.zeros(1, np.int32)
p._lib.plh_init_consistent(p._h, 1, th.ctypes.data, 0, -1.0, 1e-3, Y.ctypes.data, YP.ctypes.data, st.ctypes.data, it.ctypes.data, 0, None)
print("iters", it[0], ito, "Y maxrel", np.abs(Y-Yo).max()/np.abs(Yo).max())
print("YP diff part rel", np.abs(YP[:Nd]-YPo[:Nd]).max()/np.abs(YPo[:Nd]).max())
for a,b,nm in ((230,250,'j'),(250,280,'pe'),(280,300,'ps')):
    print(nm, "YP rel diff", np.abs(YP[a:b]-YPo[a:b]).max()/np.abs(YPo[a:b]).max())
print("YP pe dev", YP[270:280]); print("YP pe orc", YPo[270:280])
# residual at Ytmp by both
dt = 0.01
Ytmp = Yo + dt * np.concatenate([YPo[:Nd], np.zeros(N-Nd)])
YPd = np.concatenate([YPo[:Nd], np.zeros(N-Nd)])
Fo = O.residual("lco_iso", th, Ytmp, YPd, 0, -1.0)
Fd = np.zeros((1,N)); Th1 = th[None,:].copy(); Yt1 = Ytmp[None,:].copy(); YP1 = YPd[None,:].copy()
p._lib.plh_residual(p._h, 1, Th1.ctypes.data, Yt1.ctypes.data, YP1.ctypes.data, 0, -1.0, Fd.ctypes.data, 0, None)
print("R_alg(Ytmp) pe rows: orc", Fo[270:280]); print("dev", Fd[0,270:280]); print("diff", (Fd[0]-Fo)[250:280])
# solve with the oracle's LU of J at Yo (cj=0, full system): emulate FD solve approx using full J (alg block)
cp, ri, nz = O.jacobian("lco_iso", th, Yo, np.zeros(N), 0.0, 0, -1.0)
A = parity.dense_from_csc(N, cp, ri, nz)[Nd:, Nd:]
xo = parity.ld_solve(A, Fo[Nd:]); xd = parity.ld_solve(A, Fd[0, Nd:])
print("YPalg from orc R:", (-xo/dt)[20+20:20+30]); print("YPalg from dev R:", (-xd/dt)[40:50])
print("rel diff due to R difference", np.abs(xo-xd)[20:50].max()/np.abs(xo)[20:50].max())
