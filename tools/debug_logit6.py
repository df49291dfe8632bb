import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, mcmc_amd, orc
from mcmc_amd import synth
import test_gpu_parity_mala as T
def run(d, N, C, eps, burn, keep):
    X, y = synth.logistic_problem(d, N, seed=4)
    init = synth.initial_states(C, d, seed=41) * 0.1
    st = mcmc_amd.default_settings(rng_seed_value=123, n_burnin_draws=burn, n_keep_draws=keep, step_size=eps)
    g_draws, g = mcmc_amd.mala(mcmc_amd.TARGET_LOGISTIC, init, st, X=X, y=y, chain0=3)
    nb, bs = T._blocks(d)
    t = orc.TargetSpec(orc.TARGET_LOGISTIC, d, X=X, y=y, W=4, blocks=nb, block_size=bs)
    s = orc.make_settings(seed=123, n_burnin=burn, n_keep=keep, step=eps, W=4, hoist=1, blocks=nb, block_size=bs)
    o_draws, o = orc.run_many(orc.ALGO_MALA, t, init, s, chain0=3)
    return g_draws, o_draws, g, o
for c in T.CASES:
    T.test_mala_bit_exact_vs_oracle(*c)
T.test_mala_rejects_and_samples_the_target()
g, o, gi, oi = run(5, 40, 16, 0.10, 5, 20); print("d=5 ok", np.array_equal(g, o))

os.environ["MI_DEBUG_DUMP"] = "/tmp/dump.bin"
g, o, gi, oi = run(64, 100, 20, 0.05, 0, 2)
print("failing state: equal", np.array_equal(g, o))
buf = np.fromfile("/tmp/dump.bin")
NB, NSQ, NTQ = 7, 4, 1
xe = buf[:NB*4*NSQ*64].reshape(NB, 4, NSQ, 64); xg = buf[NB*4*NSQ*64:2*NB*4*NSQ*64].reshape(NB, 4, NTQ, 4, 64); yp = buf[2*NB*4*NSQ*64:]
X, y = synth.logistic_problem(64, 100, seed=4)
Xp = np.zeros((112, 64)); Xp[:100] = X; ypad = np.zeros(112); ypad[:100] = y
lane = np.arange(64)
exp_xe = np.zeros_like(xe); exp_xg = np.zeros_like(xg)
for b in range(NB):
    for q in range(4):
        for s in range(NSQ):
            exp_xe[b, q, s] = Xp[16*b + (lane & 15), q*16 + 4*s + (lane >> 4)]
        for sp in range(4):
            exp_xg[b, q, 0, sp] = Xp[16*b + 4*sp + (lane >> 4), q*16 + (lane & 15)]
print("XE ok", np.array_equal(xe, exp_xe), "XG ok", np.array_equal(xg, exp_xg), "ypad ok", np.array_equal(yp, ypad))
if not np.array_equal(yp, ypad): print("ypad got", yp[:32], "\n want", ypad[:32])
if not np.array_equal(xe, exp_xe): print("XE mismatch blocks", np.nonzero(np.abs(xe-exp_xe).max(axis=(1,2,3)))[0])
bad = np.abs(xe - exp_xe) > 0
print("mismatch count per block", bad.sum(axis=(1,2,3)), "per q (block0)", bad[0].sum(axis=(1,2)), "per s (block0,q0)", bad[0,0].sum(axis=1))
X5, y5 = synth.logistic_problem(5, 40, seed=4)
X5p = np.zeros((48, 64)); X5p[:40, :5] = X5
old_xe = np.zeros((3, 4, 4, 64))
for b in range(3):
    for q in range(4):
        for s in range(4):
            old_xe[b, q, s] = X5p[16*b + (lane & 15), q*16 + 4*s + (lane >> 4)]
print("dumped blocks 0..2 equal OLD call's XE:", np.array_equal(xe[:3], old_xe), " fraction of mismatching entries equal to old:", (xe[:3][bad[:3]] == old_xe[bad[:3]]).mean())
print("yp mismatch idx", np.nonzero(yp != ypad)[0], "old y at those", [ (y5[i] if i < 40 else 0.0) for i in np.nonzero(yp != ypad)[0]])
np.set_printoptions(precision=4, linewidth=200)
print("XE[0,0,0,:8] got", xe[0,0,0,:8]); print("           want", exp_xe[0,0,0,:8]); print("            old", old_xe[0,0,0,:8])
print("XE[1,2,1,:8] got", xe[1,2,1,:8]); print("           want", exp_xe[1,2,1,:8])
# is the got data a valid fragment of the NEW X at some other (row,col)?
val = xe[1,2,1,5]
print("value", val, "found in new X at", np.argwhere(X == val)[:4], "in old X5 at", np.argwhere(X5 == val)[:4])
