set -e
OUT=gpurun_out/final_r3c
mkdir -p $OUT
R=$PWD
python tools/hazard_repro_backward.py > $OUT/hazard_repro_backward_default_build.txt 2>&1
GA_VARIANT_LIB=garage_amd/_C/variants/lib_slp.so python tools/hazard_repro_backward.py > $OUT/hazard_repro_backward_slp_build.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_no -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-envs 0 --no-overlap --no-split-variant > $R/$OUT/prof_no.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_ov -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-envs 0 --no-split-variant > $R/$OUT/prof_ov.log 2>&1
GARAGE_AMD_SPLIT_BF16=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_split_no -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-envs 0 --no-overlap > $R/$OUT/prof_split_no.log 2>&1
echo kernel stats done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-envs 0 --no-roofline --no-overlap --no-split-variant > $R/$OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-envs 0 --no-roofline --no-overlap --no-split-variant > $R/$OUT/pmc_write.log 2>&1
GARAGE_AMD_SPLIT_BF16=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_split_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-envs 0 --no-roofline --no-overlap > $R/$OUT/pmc_split_fetch.log 2>&1
GARAGE_AMD_SPLIT_BF16=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pmc_split_write -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-envs 0 --no-roofline --no-overlap > $R/$OUT/pmc_split_write.log 2>&1
cd $R
find $OUT -name '*_kernel_trace.csv' -delete
find $OUT -name '*_agent_info.csv' -delete
ls $OUT
