#!/bin/bash
# tools/stagger_ab.sh lib1.so ... : lockstep / staggered c3 throughput of alternative builds on one box
for L in default "$@"; do
  if [ $L = default ]; then python tools/stagger_experiment.py 2>/dev/null | tail -n 2 | sed "s|^|$L |"
  else PCBENV_LIB=$GRAFT_REPO_ROOT/$L python tools/stagger_experiment.py 2>/dev/null | tail -n 2 | sed "s|^|$L |"; fi
done
