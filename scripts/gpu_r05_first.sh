#!/bin/bash
# First GPU trip of the NEXT round (prepared at the end of round 4, when the builder's GPU budget was spent): what was built but not
# measured.  ~1.5 min of box time.
#   1. pgt_tune("seq_vdot", 1): the gate products of the one-workgroup DCRNN sequences one thread per node (csrc/seq_small.hip,
#      sq_dot_node) — bit-identical on the CPU test double; expected from the phase timeline (profiles/r04o_*): forward step 14.0 ->
#      ~10 us, backward 25.0 -> ~19 us.  If B = 64 AND B = 1024 both gain, make it the default (g_seq_vdot = 1); the VDOT backward
#      kernel needs 151 VGPRs (one workgroup per CU), so B = 1024 may lose — then switch by launch size.
#   2. the phase timeline again (lab/seq_small_lab is built by scripts/build_lab.sh).
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for B in 64 256 1024; do
  for V in 0 1; do
    echo -n "seq_vdot=$V "; PGT_TUNE=seq_vdot=$V timeout 100 python scripts/small_batch_probe.py $B 2 100 2>&1 | tail -1
  done
done
timeout 60 python -m pytest tests/test_dcrnn.py -m gpu -q -k "one_workgroup" 2>&1 | tail -2
[ -x lab/seq_small_lab ] && timeout 20 ./lab/seq_small_lab 64 2 2 3 | head -8
