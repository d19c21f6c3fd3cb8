for r in 1 2 3; do
for v in head.so ""; do
  if [ -n "$v" ]; then export NMPC_HIP_DDP_LIB=/root/repo/nmpc_amd/lib/alt/$v; else unset NMPC_HIP_DDP_LIB; fi
  echo -n "${v:-current}: "; python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['roofline'].get('kernel_ms'))"
done; done
