#!/bin/bash
# kRateDrop (the PCG-rate trigger of the lazy refresh) as a compile-time variant: product builds with -DUZL_RATE_DROP=x kept beside the
# library, swept over the large and the small shape sets
for rd in default 0.4 0.5 0.75 0.9; do
  if [ $rd = default ]; then unset UZL_LIB; else export UZL_LIB=$PWD/uzliti_slam_amd/libuzl_ab_rd$rd.so; fi
  for set in large small; do
    python3 tests/diag/refresh_shapes.py $([ $set = small ] && echo small) 2>&1 | python3 -c "
import sys
ms=[]; pcg=0
for ln in sys.stdin:
    p=ln.split()
    if 'seed' in p and 'ms' in p: ms.append(float(p[p.index('ms')-1])); pcg+=int(p[p.index('pcg')+1])
print('rate drop %-8s %-6s' % ('$rd', '$set'), ' '.join('%7.1f' % m for m in ms), ' | sum %.1f (without the last %.1f)  pcg %d' % (sum(ms), sum(ms[:-1]), pcg))"
  done
done
