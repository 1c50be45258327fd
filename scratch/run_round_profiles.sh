#!/bin/bash
# Round-end measurement recipe (run through gpurun from the repo root).
set -x
R=$PWD
mkdir -p $R/gpurun_out/final
python bench.py > $R/gpurun_out/final/bench_default.json 2> $R/gpurun_out/final/bench_default.err
tail -c 3000 $R/gpurun_out/final/bench_default.json
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final/prof4s -o r -- python $R/bench.py --no-cpu-baseline --no-cobatch-extra > $R/gpurun_out/final/prof4s.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final/prof1s -o r -- python $R/bench.py --no-cpu-baseline --no-cobatch-extra --streams 1 > $R/gpurun_out/final/prof1s.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/final/pmc_fetch -o r -- python $R/bench.py --no-cpu-baseline --no-cobatch-extra --streams 1 --steps 20 --warmup 5 > $R/gpurun_out/final/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/final/pmc_write -o r -- python $R/bench.py --no-cpu-baseline --no-cobatch-extra --streams 1 --steps 20 --warmup 5 > $R/gpurun_out/final/pmc_write.log 2>&1
cd $R
ls -la gpurun_out/final gpurun_out/final/*
du -sh gpurun_out/final
