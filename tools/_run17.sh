export TMPDIR=/tmp
O=gpurun_out/r04p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_composites.py tests/test_gpu_encode.py tests/test_gpu_parallel.py -x -q > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for i in 1 2 3; do python tools/lt_direct_probe.py 512 10; done
python bench.py --steps 20 --cpu-seconds 0 --variant-keys 0 --stream-keys 0 --key-per-item 0 | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value']), json.dumps(d['lt_sharded'])[:700])"
