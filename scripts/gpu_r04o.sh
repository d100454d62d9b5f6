#!/bin/bash
# Round 4, trip o: phase timeline of the one-workgroup sequence kernels (lab/seq_small_lab), nothing else
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
(timeout 20 ./lab/seq_small_lab 64 2 2 3; timeout 20 ./lab/seq_small_lab 64 2 8 3 | head -8) > gpurun_out/seq_small_lab.txt 2>&1
cat gpurun_out/seq_small_lab.txt
