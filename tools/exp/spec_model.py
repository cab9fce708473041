"""EXPERIMENTS.md R6.6 — would SPECULATIVE simulation hide C2's serial chain?  A model, not a kernel.

The chain kernel's iteration is  [wait for the cone's publications 1.5 | walk 1.6 | donor's record 0.75 | proposal 0.35 | simulation 6.65 | accept + publish 1.1]  (us,
profiles/r06_phase_stamps.txt).  A chain that is NOT exchanged continues from its own record, so its next proposal is known at its own accept step: the workers could
simulate such chains while the control wave waits, walks and reads.  Model: 256 tiles, every tile's walk needs the publications of ~100 others (its cone); pass A =
8 chains simulated speculatively right behind the publication, pass B = the other 8 behind the walk, then the mispredicted chains of pass A again (a pass of 1 / 2 /
4 / 8 accumulators), then the accept step.  An iteration lasts as long as the slowest tile of a cone: the re-simulation's TAIL is what decides.
  python tools/exp/spec_model.py"""
import numpy as np

rng = np.random.default_rng(1)
NT, CONE, IT = 256, 100, 400


def nch(m):
    return 0 if m == 0 else 1 if m == 1 else 2 if m == 2 else 4 if m <= 4 else 8


def run(policy, p_ex=0.26, per_pair=None):
    f = np.zeros(NT)
    C1, PA, BOOK, LV, WALK, PROP, PB, ACC = 0.35, 3.55, 0.5, 1.5, 1.6, 0.5, 3.55, 1.1
    hist = []
    for t in range(IT):
        cones = [rng.choice(NT, CONE, replace=False) for _ in range(NT)]
        if policy == "now":
            nf = np.array([max(f[x], f[cones[x]].max() + LV) + WALK + 0.75 + 0.35 + 6.65 + 1.12 for x in range(NT)])
        else:
            if policy == "first8":      # the tile's first eight chains, whatever the plan says about them
                m = rng.binomial(8, p_ex, NT)
            else:                        # the eight chains of the tile with the fewest pairs in this exchange (Poisson(2) pairs per chain)
                k = rng.poisson(2.0, (NT, 16)); k.sort(axis=1)
                m = (rng.random((NT, 8)) < 1 - (1 - per_pair) ** k[:, :8]).sum(axis=1)
            rs = np.array([0.0 if mm == 0 else 0.4 + 0.45 * nch(mm) for mm in m])
            nf = np.empty(NT)
            for x in range(NT):
                a = f[x] + C1 + PA
                b = max(f[x] + BOOK, f[cones[x]].max() + LV) + WALK + PROP
                nf[x] = max(a, b) + PB + rs[x] + ACC
        hist.append(nf.mean()); f = nf
    h = np.array(hist)
    return (h[-1] - h[100]) / (IT - 1 - 100)


print("the kernel as it is                                              %.2f us per iteration" % run("now"))
print("speculate on the first 8 chains, 26 %% of the chains exchanged    %.2f" % run("first8"))
print("speculate on the first 8 chains,  3 %% exchanged (min_improve 0.5) %.2f" % run("first8", 0.03))
for q in (0.08, 0.12, 0.15):
    k = rng.poisson(2.0, 100000)
    print("speculate on the 8 chains with the fewest pairs, a pair swaps with probability %.2f (%.0f %% of the chains exchanged)   %.2f"
          % (q, 100 * (1 - (1 - q) ** k).mean(), run("least8", per_pair=q)))
