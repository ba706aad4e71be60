#!/bin/bash
# GPU box: the few-channel layers' product kernels in their inference and training forms (tools/few_channel_bench.py), launch time over
# consecutive blocks of launches; then the stamps of the tap-row kernel with and without the channel sums
cd /tmp
for m in ${*:-first first_exact bwd9 bwd9_sums fwd5}; do RISP_FCB_SERIES=1 python3 $GRAFT_REPO_ROOT/tools/few_channel_bench.py $m 32 256 256 8 10 2>&1 | tail -2; done
