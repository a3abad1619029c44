#!/bin/bash
# On the GPU box: tools/time_wide.py under the stock library and under every build in graphsage_amd/variants.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT; export PYTHONPATH=$ROOT
echo "stock: $(timeout -k 10 120 python3 tools/time_wide.py "$@" | tail -1)" || exit 1
for so in graphsage_amd/variants/libgsage_*.so; do
  out=$(GSAGE_LIB=$ROOT/$so timeout -k 10 120 python3 tools/time_wide.py "$@" | tail -1) || { echo "$so failed"; exit 1; }
  echo "$(basename $so): $out"
done
