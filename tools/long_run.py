"""Does the step time drift over a long run? bench.py's step in one process, per-window means of HIP-event step times (a leak in an event
pool or workspace ring would show as a slope; DVFS shows as a plateau after the first seconds).  python tools/long_run.py [steps] [window]"""
import os, subprocess, sys, json
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
win = int(sys.argv[2]) if len(sys.argv) > 2 else 100
os.environ['PB_LONG_RUN_DUMP'] = '/tmp/pb_long_run.json'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', str(steps), '--warmup', '5', '--no-cpu-baseline', '--no-probe'], capture_output=True, text=True)
d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
t = json.load(open('/tmp/pb_long_run.json'))
print('mean %.2f ms/step over %d steps' % (d['ms_per_step'], steps))
print('per-%d-step windows (ms):' % win, ' '.join('%.2f' % (sum(t[i:i + win]) / len(t[i:i + win])) for i in range(0, len(t), win)))
