cd /root/repo
for st in 20 100 500 2500; do
  timeout -k 10 200 python bench.py --lib $PWD/variants/v15g.so --steps $st --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('steps $st', d['ms_per_step'], 'ms', d['value'], 'GB/s frac', d['roofline']['frac'])"
done
timeout -k 10 200 python bench.py --lib $PWD/variants/v15g.so --steps 600 --warmup 3 --no-cpu-baseline --gib-per-gpu 3.9 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('3.9 GiB steps 600', d['ms_per_step'], 'ms', d['value'], 'GB/s frac', d['roofline']['frac'])"
timeout -k 10 200 python bench.py --lib $PWD/variants/v15g.so --steps 20 --warmup 3 --no-cpu-baseline --gib-per-gpu 3.9 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('3.9 GiB steps 20', d['ms_per_step'], 'ms', d['value'], 'GB/s frac', d['roofline']['frac'])"
