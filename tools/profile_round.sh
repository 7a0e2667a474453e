#!/bin/bash
# end-of-round evidence: bench lines (default / fp32 UNet / SSIM), rocprofv3 kernel-trace summaries (two decoder streams and
# single stream), PMC traffic.  Everything lands under gpurun_out/; copy what is to be judged into profiles/.
#   tools/profile_round.sh r02_g
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PMC_TAG=${PMC_TAG:-r04} bash tools/pmc_traffic.sh          # first: bench.py quotes roofline.traffic from the summary of THIS build (csrc hash)
python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
python3 bench.py --model unet --precision fp32 --batch 16 --no-cpu-baseline > gpurun_out/${tag}_bench_unet_fp32.json 2>> gpurun_out/${tag}_bench.err
python3 bench.py --ssim --no-cpu-baseline > gpurun_out/${tag}_bench_ssim.json 2>> gpurun_out/${tag}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python3 bench.py --steps 20 --warmup 5 --blocks 1 --no-cpu-baseline > gpurun_out/${tag}_profiled_bench.json 2> gpurun_out/${tag}_prof.err
UNET_TWO_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof1 -- python3 bench.py --steps 20 --warmup 5 --blocks 1 --no-cpu-baseline > gpurun_out/${tag}_profiled1_bench.json 2> gpurun_out/${tag}_prof1.err
find gpurun_out/${tag}_prof gpurun_out/${tag}_prof1 -name "*kernel_stats.csv" | while read f; do cp $f gpurun_out/$(echo $f | cut -d/ -f2)_kernel_stats.csv; done
find gpurun_out/${tag}_prof gpurun_out/${tag}_prof1 -name "*kernel_trace.csv" -delete
ls gpurun_out | grep ${tag}
