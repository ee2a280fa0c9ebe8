#!/bin/bash
# usage: perf/shapes.sh [label]   -- microbench of the Llama-8B tcomb_6_7 shapes (graph replay)
for s in "14336 4096" "4096 14336" "4096 4096" "1024 4096" "28672 4096" "6144 4096"; do
  set -- $s
  timeout -k 10 120 python perf/microbench.py --m $1 --k $2 --iters 10 --graph 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%6d x %6d  %7.2f us  %7.1f GB/s' % (d['m'], d['k'], d['us_per_launch'], d['GBps']))"
done
