#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=gpurun_out
( time timeout 1700 python -m pytest tests -q -m gpu -s ) > $O/gputest_full.log 2>&1; tail -4 $O/gputest_full.log
grep "^\[parity\]\|^\[fp16s\|^\[post-processor\|passed\|failed\|default route" $O/gputest_full.log > $O/gputest_parity_deltas.log
