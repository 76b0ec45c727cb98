for rep in 1 2 3; do for ps in 1 3 6; do
python3 bench.py --steps 20 --warmup 5 --no_cpu_baseline --prime_seconds $ps 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prime $ps: value %.1f sustained %.1f' % (d['value'], d['sustained']['value']))"
done; done
