#!/bin/bash
# one-variable sweeps of the solver's diagnostic switches over tests/diag/refresh_shapes.py's graph shapes (diagnostic build): one row
# per setting - solve times in ms per shape, their sum, PCG iterations in total.   [SHAPES=small] bash tests/diag/knob_sweep.sh "UZL_X=1" "UZL_X=2" ...
export UZL_LIB=$PWD/uzliti_slam_amd/libuzl_mi355x_diag.so
for kv in "$@"; do
  env $kv python3 tests/diag/refresh_shapes.py $SHAPES 2>&1 | python3 -c "
import sys
ms=[]; pcg=0
for ln in sys.stdin:
    p=ln.split()
    if 'seed' in p and 'ms' in p: ms.append(float(p[p.index('ms')-1])); pcg+=int(p[p.index('pcg')+1])
print('%-34s' % '$kv', ' '.join('%7.1f' % m for m in ms), ' | sum %.1f (without the last %.1f)  pcg %d' % (sum(ms), sum(ms[:-1]), pcg))"
done
