#!/bin/bash
cd $GRAFT_REPO_ROOT
for occ in 12 16 24 32; do for fb in 0.6 0.8 1.0 1.2; do
  echo "occ=$occ first=$fb: $(COFI_KNN_OCC=$occ COFI_KNN_FIRST=$fb GMIN=1024 python scratch/knn_time.py 2>&1 | grep -E '^sum|self0|self4' | tr '\n' ' ' | sed 's/brute [0-9.]* ms (+ 5 grid builds)//')"
done; done
