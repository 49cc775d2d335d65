#!/bin/bash
# round 2: the whole -m gpu suite, then the default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02b
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1
rc=$?
echo "tests rc=$rc"
tail -8 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"
tail -3 $O/bench.err
cat $O/bench.json
