#!/bin/bash
# Measurement pass on the GPU box: every bench line DESIGN.md quotes, the rocprofv3 kernel statistics and the two PMC passes of
# the headline command.  Writes gpurun_out/<tag>/ (scratch) and, with COPY=1 (default), the summaries into profiles/<round>_*.
#   gpurun --timeout 1800 -- 'bash tools/bench_all.sh r06'
# Every line is produced by bench.py itself (one JSON object per file), so a claim in DESIGN.md can be re-run verbatim.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=${1:-r06}; O=gpurun_out/$R; mkdir -p $O profiles
run() { out=$1; shift; python3 bench.py "$@" > $O/$out.json 2> $O/$out.err || echo "FAILED: $out" >&2; }
run bench_train                                                                    # BASELINE configs[1], the driver's command
run bench_train_metrics_on --metrics 1 --no-cpu-baseline
run bench_infer_b64_hipgraph --mode infer --batch 64 --steps 20 --warmup 5         # configs[4]: tiles/s + p50 batch latency
run bench_train_512_b4 --img 512 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline # configs[3], 1 GPU
run bench_embed_b64 --mode embed --batch 64 --steps 20 --warmup 5                  # SURVEY 8f row 4 (fp16 operands: the reference's .half())
run bench_infer_b64_hipgraph_fp16 --mode infer --batch 64 --steps 20 --warmup 5 --dtype fp16   # configs[4] under generator.eval().cuda().half()
run bench_unetr_train --generator unet_lora --no-cpu-baseline --steps 10 --warmup 3
run bench_unetr_infer_b64 --generator unet_lora --mode infer --batch 64 --steps 10 --warmup 3
MIPHEI_FORCE_DDP=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 \
    > $O/bench_train_rccl_1rank.json 2> $O/bench_train_rccl_1rank.err              # the bucketed exchange on one rank
rocprofv3 --kernel-trace --stats -d $O/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --comm-standin 0 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --comm-standin 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --comm-standin 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --comm-standin 0 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --comm-standin 0 > /dev/null 2>&1
python3 tools/step_counters.py $O/trace $O/pmc_sq $O/pmc_fetch $O/pmc_write > $O/step_counters.txt 2> $O/step_counters.err
db=$(ls $O/stats/*/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python3 tools/prof_summary.py $db 70 > $O/kernel_stats_train.txt
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json > $O/pmc_traffic.txt
# the same kernel table + per-kernel counters for the other two single-GPU configurations (round 6): configs[4] (batch-64 inference; eager, so
# that the table carries every launch of ONE forward per step -- the graph replays the same kernels) and configs[3] (512 x 512, B = 4)
profile_cfg() { tag=$1; shift
  rocprofv3 --kernel-trace --stats -d $O/stats_$tag -- python3 bench.py "$@" > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$tag -- python3 bench.py "$@" --steps 2 --warmup 1 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$tag -- python3 bench.py "$@" --steps 2 --warmup 1 > /dev/null 2>&1
  rocprofv3 --kernel-trace --output-format csv -d $O/trace_$tag -- python3 bench.py "$@" --steps 2 --warmup 1 > /dev/null 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq_$tag -- python3 bench.py "$@" --steps 2 --warmup 1 > /dev/null 2>&1
  python3 tools/step_counters.py $O/trace_$tag $O/pmc_sq_$tag $O/pmc_fetch_$tag $O/pmc_write_$tag > $O/step_counters_$tag.txt 2> $O/step_counters_$tag.err
  db=$(ls $O/stats_$tag/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python3 tools/prof_summary.py $db 40 > $O/kernel_stats_$tag.txt
  rm -rf $O/stats_$tag $O/pmc_fetch_$tag $O/pmc_write_$tag $O/trace_$tag $O/pmc_sq_$tag
}
profile_cfg infer_b64 --mode infer --batch 64 --graph 0 --probe 0 --steps 6 --warmup 2
profile_cfg train_512_b4 --img 512 --batch 4 --no-cpu-baseline --comm-standin 0 --probe 0 --steps 6 --warmup 2
# SQ counter breakdown of the GEMM / attention / LayerNorm kernels inside the step (six more --pmc passes)
bash tools/sq_counters.sh $O/gemm_sq_counters.txt gemm_ws_kernel gemm_kernel attn_ ln_fwd_lora ln_bwd > /dev/null 2>&1
# micro-benchmarks quoted in DESIGN.md: GEMMs with their epilogues vs hipBLASLt, attention, decoder convolutions, small kernels
python3 tools/bench_epi.py > $O/gemm_epilogues.txt 2>/dev/null
python3 tools/bench_vs_blas.py > $O/gemm_vs_hipblaslt.txt 2>/dev/null
{ python3 tools/bench_attn.py 329 ours; python3 tools/bench_attn.py 1301 ours; } > $O/attn.txt 2>/dev/null
python3 tools/bench_decoder_convs.py > $O/decoder_convs.txt 2>/dev/null
python3 tools/bench_small.py > $O/small_kernels.txt 2>/dev/null
python3 tools/bench_lora_wgrad.py > $O/lora_wgrad.txt 2>/dev/null
# where a wave-specialised GEMM launch's time goes (timing build: make DEBUG_KNOBS=1 BUILD=build_tm LIB=variants/libmiphei_tm.so EXTRA=-DMVIT_WS_TIMING)
[ -f miphei-vit_amd/csrc/variants/libmiphei_tm.so ] && MIPHEI_LIB=miphei-vit_amd/csrc/variants/libmiphei_tm.so python3 tools/ws_timing.py 2>/dev/null | grep -v amdgpu.ids > $O/ws_timing.txt
python3 tools/gemm_power.py 2>/dev/null | grep -v amdgpu.ids > $O/gemm_power_8k.txt
# run-to-run identity of the whole step (deterministic mode, 1000 repeats of forward + loss + backward on one input, bitwise) and of the
# hipGraph-free inference forward at batch 64
{ MIPHEI_DETERMINISTIC=1 python3 tools/debug/step_soak.py 1000 16 256 myvitmatte train 2>/dev/null | tail -2
  MIPHEI_DETERMINISTIC=1 python3 tools/debug/step_soak.py 500 64 256 myvitmatte infer 2>/dev/null | tail -1; } > $O/step_soak.txt
if [ "${COPY:-1}" = 1 ]; then
  for f in $O/bench_*.json $O/kernel_stats_*.txt $O/pmc_traffic.json $O/step_counters*.txt $O/gemm_sq_counters.txt $O/gemm_epilogues.txt \
           $O/gemm_vs_hipblaslt.txt $O/attn.txt $O/decoder_convs.txt $O/small_kernels.txt $O/lora_wgrad.txt $O/step_soak.txt $O/ws_timing.txt; do
    [ -s "$f" ] && cp $f profiles/${R}_$(basename $f)
  done
fi
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/trace $O/pmc_sq      # raw traces are large; the summaries above are what is kept
for f in $O/bench_*.json; do echo "== $f"; cut -c1-400 $f; done; head -25 $O/kernel_stats_train.txt; cat $O/pmc_traffic.txt
