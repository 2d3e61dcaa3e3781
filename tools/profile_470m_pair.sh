set -e
repo=$PWD
for q in mixed_precision none; do
  out=$repo/gpurun_out/prof_470m_$q; mkdir -p $out
  (cd /tmp && TMPDIR=/tmp timeout -k 10 280 rocprofv3 --kernel-trace --stats -d $out -o run --output-format csv -- python3 $repo/train_bench.py --model 470m --quantize $q --compile --steps 6 --warmup 3 > $out/stdout.log 2> $out/rocprof.log)
  tail -2 $out/stdout.log
  python3 tools/train_profile_summary.py $out/run_kernel_trace.csv 2 > $out/summary.txt
  head -40 $out/summary.txt
done
