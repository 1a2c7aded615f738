#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r5full; rm -rf $O; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -n 25 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
