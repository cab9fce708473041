#!/bin/bash
# tools/exp/shard_bytes.sh: what a shard in the persistent form WRITES per iteration (VERDICT r5 "Next #2a"): the ranks as 8 / 4 / 2 processes on the one
# GPU (HIP IPC: every window is this device's memory, so WRITE_SIZE counts the stores into ALL windows), rocprofv3 --pmc WRITE_SIZE per process,
# k_chain_persist_loc<2, false, true> only; the library of round 5 (tools/exp/libsmmhip_old.so: 8-byte slot + the whole record into every window)
# against this round's (parameters + value into the peers', the whole record into the own window)
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/shard_bytes; mkdir -p $out
export TMPDIR=/tmp
for v in old new; do
  cp tools/exp/libsmmhip_$v.so smm.jl_amd/csrc/libsmmhip.so
  for G in ${SB_G:-8 2}; do
    rm -rf /tmp/sb && (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/sb -- python $GRAFT_REPO_ROOT/tools/exp/sharded_persist_time.py $G > $out/run_${v}_$G.txt 2>&1)
    python - "$v" "$G" <<'PY' >> $out/summary.txt
import csv, glob, sys
v, G = sys.argv[1], int(sys.argv[2])
tot, launches, procs = 0.0, 0, 0
for f in glob.glob("/tmp/sb/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "k_chain_persist_loc" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE"]
    if rows:
        procs += 1; launches += len(rows); tot += sum(float(r["Counter_Value"]) for r in rows)
iters = 801   # per process: 1 + 200 warm-up + 3 x 200 (tools/exp/sharded_persist_time.py); the first is a per-iteration launch
if procs:
    per_rank_iter = tot * 1024.0 / procs / (iters - 1)
    chains = 4096 // G
    print("%s library, %d ranks x %d chains: WRITE_SIZE of k_chain_persist_loc<2, false, true> = %.1f KB per rank and iteration (%d launches in %d processes) = %.0f B per chain and iteration"
          % (v, G, chains, per_rank_iter / 1024.0, launches, procs, per_rank_iter / chains))
else:
    print("%s library, %d ranks: no counter rows found" % (v, G))
PY
  done
done
cp tools/exp/libsmmhip_new.so smm.jl_amd/csrc/libsmmhip.so
cat $out/summary.txt
