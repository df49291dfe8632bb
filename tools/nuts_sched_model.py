"""List-scheduling model of configs[3]'s NUTS run from the per-chain leapfrog totals of a real run.

    python tools/nuts_sched_model.py n_leap.npy [n_leap_burnin.npy n_leap_kept.npy eps.npy]  > summary.json

n_leap.npy: int array [C], mi_chains.n_leapfrogs of one full-size run (any nuts kernel: they execute the same leapfrogs).  The model
assumes a constant tick time: a wave of the static kernel (nuts_reg.hpp) runs max over its 16 chains ticks, a workgroup of four waves
holds its CU until its slowest wave is done, workgroups are list-scheduled on the CUs in index order; the dynamic kernel (nuts_dyn.hpp)
list-schedules CHAINS on chain slots."""
import heapq, json, sys
import numpy as np


def listsched(jobs, m):
    h = [0.0] * m
    heapq.heapify(h)
    for j in jobs:
        heapq.heappush(h, heapq.heappop(h) + j)
    return max(h)


def main():
    n = np.load(sys.argv[1]).astype(np.float64)
    C, n_cu = n.size, 256
    slots = 64 * n_cu
    w = n[: C // 16 * 16].reshape(-1, 16).max(axis=1)
    g = n[: C // 64 * 64].reshape(-1, 64).max(axis=1)
    ideal = n.sum() / slots
    out = {"chains": int(C), "n_leap": {"mean": n.mean(), "std": n.std(), "min": n.min(), "max": n.max(),
                                        "quantiles_1_10_50_90_99": np.quantile(n, [0.01, 0.1, 0.5, 0.9, 0.99]).tolist()},
           "static": {"lane_efficiency_in_a_wave_mean_over_max16": n.mean() / w.mean(),
                      "wave_efficiency_in_a_workgroup": w.mean() / g.mean(), "combined": n.mean() / g.mean(),
                      "makespan_ticks": listsched(g, n_cu), "over_balanced": listsched(g, n_cu) / ideal},
           "dynamic_per_slot": {"makespan_ticks": listsched(n, slots), "over_balanced": listsched(n, slots) / ideal},
           "balanced_ticks": ideal}
    out["model_speedup_dynamic_over_static"] = out["static"]["makespan_ticks"] / out["dynamic_per_slot"]["makespan_ticks"]
    if len(sys.argv) >= 5:
        n1, n2, eps = (np.load(a).astype(np.float64) for a in sys.argv[2:5])
        out["burn_in_half"] = {"mean": n1.mean(), "std": n1.std(), "max": n1.max()}
        out["kept_half"] = {"mean": n2.mean(), "std": n2.std(), "max": n2.max(), "corr_with_inverse_step_size": float(np.corrcoef(n2, 1 / eps)[0, 1])}
        a, b = listsched(n1, slots), listsched(n2[np.argsort(eps)], slots)
        out["two_launches_second_ordered_by_step_size"] = {"makespan_ticks": a + b, "over_balanced": (a + b) / ideal}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
