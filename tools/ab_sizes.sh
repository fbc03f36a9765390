#!/bin/bash
# several builds of the library (files under ab/) at several batch sizes on one box:   tools/ab_sizes.sh "A.so B.so" REPS "R1xE1 R2xE2 ..."
LIBS=$1; REPS=${2:-2}; SIZES=$3
cd $GRAFT_REPO_ROOT
for sz in $SIZES; do
  R=${sz%x*}; E=${sz#*x}
  echo "== $R regions x $E cost weights"
  bash tools/ab_variants.sh "$LIBS" $REPS --regions $R --eps $E
done
