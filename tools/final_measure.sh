#!/bin/bash
# Developer tool: the round's measurement batch on one MI355X box (gpurun):
# benches of every config, rocprofv3 kernel stats and the two PMC passes.
# Usage (on the GPU box, from the repo root): bash tools/final_measure.sh <outdir>
set -e
OUT=${1:-gpurun_out/final}
mkdir -p $OUT
R=$PWD
python bench.py > $OUT/c3.log 2> $OUT/c3.err
# run-to-run spread of the headline (five consecutive runs, one box)
for i in 1 2 3 4 5; do python bench.py --cpu-envs 0 --no-scan-c4 --no-roofline > $OUT/c3_rep$i.log 2>&1; done
python bench.py --no-overlap --cpu-envs 0 > $OUT/c3_serial.log 2>&1
# the data-parallel code path priced on one GPU (1-rank RCCL group)
python bench.py --dp --cpu-envs 0 --no-scan-c4 > $OUT/c3_dp.log 2>&1
python bench.py --dp --no-overlap --cpu-envs 0 --no-scan-c4 > $OUT/c3_dp_serial.log 2>&1
GARAGE_AMD_MERGED_PAIR=1 python bench.py --cpu-envs 0 --no-scan-c4 > $OUT/c3_merged.log 2>&1
python bench.py --config c2 > $OUT/c2.log 2>&1
python bench.py --config c5 --steps 5 --warmup 2 > $OUT/c5.log 2>&1
python bench.py --config c1 > $OUT/c1.log 2>&1
python bench.py --config c3mb64 --steps 3 --warmup 1 > $OUT/c3mb64.log 2>&1
python bench.py --algo trpo > $OUT/trpo.log 2>&1
python bench.py --config c3scan > $OUT/c3scan.log 2>&1
python bench.py --config c4scan > $OUT/c4scan.log 2>&1
# the opt-in split-operand experiment: whole runs in that mode, its error against fp64,
# the two microbenchmarks behind it and the hazard seen from the library
GARAGE_AMD_SPLIT_BF16=1 python bench.py --cpu-envs 0 --no-scan-c4 > $OUT/c3_split.log 2>&1
GARAGE_AMD_SPLIT_BF16=1 python bench.py --no-overlap --cpu-envs 0 --no-scan-c4 > $OUT/c3_split_serial.log 2>&1
python tools/split_error_histogram.py > $OUT/split_error_histogram.json 2> $OUT/split_error_histogram.err
garage_amd/_C/mfma_valu_overlap > $OUT/mfma_valu_overlap.txt 2>&1
garage_amd/_C/mfma_valu_hazard > $OUT/mfma_valu_hazard.txt 2>&1
python tools/hazard_repro_backward.py > $OUT/hazard_repro_backward_default_build.txt 2>&1
GA_VARIANT_LIB=garage_amd/_C/variants/lib_slp.so python tools/hazard_repro_backward.py > $OUT/hazard_repro_backward_slp_build.txt 2>&1
echo benches done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_no -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-envs 0 --no-overlap --no-split-variant > $R/$OUT/prof_no.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_ov -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-envs 0 --no-split-variant > $R/$OUT/prof_ov.log 2>&1
GARAGE_AMD_SPLIT_BF16=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_split_no -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-envs 0 --no-overlap > $R/$OUT/prof_split_no.log 2>&1
echo kernel stats done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-envs 0 --no-roofline --no-overlap --no-split-variant > $R/$OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-envs 0 --no-roofline --no-overlap --no-split-variant > $R/$OUT/pmc_write.log 2>&1
cd $R
# keep only what is small enough to travel back
find $OUT -name '*_kernel_trace.csv' -delete
find $OUT -name '*_agent_info.csv' -delete
ls -la $OUT $OUT/*/* | head -40
