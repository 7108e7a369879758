"""Helper: run bench.py with the given args and print a one-line summary."""
import json, subprocess, sys, os
out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py')] + sys.argv[1:],
                     capture_output=True, text=True)
line = [l for l in out.stdout.splitlines() if l.startswith('{')]
if not line:
    print(out.stdout[-2000:], out.stderr[-2000:]); sys.exit(1)
d = json.loads(line[-1])
print(' | '.join('%s %.1f' % (k['kernel'][:28], k['us_per_frame']) for k in d.get('kernels', [])))
print('fps %.1f ms/step %.3f stages %s' % (d['value'], d['ms_per_step'], {k: round(v, 3) for k, v in d['stages_ms'].items()}), os.environ.get('OJF_CONV_MT') or '')
