#!/bin/bash
# usage (GPU box, repo root): tools/prof_zvb.sh <tag> <B> [<B> ...]   rocprofv3 kernel-trace statistics of tools/prof_zvb.py
tag=$1; shift
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
case "$MH_SPEC_DIR" in ""|/*) ;; *) export MH_SPEC_DIR=$root/$MH_SPEC_DIR ;; esac
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/${tag}_trace -o zvb --output-format csv -- python3 $root/tools/prof_zvb.py "$@" > $out/${tag}.log 2>&1
cd $root
f=$(find $out/${tag}_trace -name '*kernel_stats.csv' | head -1)
cp "$f" $out/${tag}_kernel_stats.csv
cut -c1-200 $out/${tag}_kernel_stats.csv | head -12
