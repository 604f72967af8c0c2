#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TSGPU_HOST_TIMING=1 timeout 600 python bench.py --workload hybrid --no-cpu-baseline 2>&1 | grep -E "tsgpu\]" | tail -4 | cut -c1-300
