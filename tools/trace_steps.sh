# tools/trace_steps.sh OUT CMD...: duration of every primal / dual sweep of CMD, in launch order (rocprofv3 --kernel-trace)
out=$1; shift
mkdir -p $out && out=$(cd $out && pwd)
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace -d $out -o t --output-format csv -- "$@" > $out/cmd.out 2> $out/cmd.err || exit 1
python3 - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
d = [(r['Kernel_Name'][5:13], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows if r['Kernel_Name'].startswith(('void k_dual4<', 'void k_primal4<'))]
with open(out + '/per_step.txt', 'w') as fh:
    for i in range(0, len(d) - 1, 2):
        p, q = d[i], d[i + 1]
        fh.write("%3d %s %.4f  %s %.4f  gap %.4f  since previous %.3f ms\n" % (i // 2, p[0], (p[2] - p[1]) / 1e6, q[0], (q[2] - q[1]) / 1e6, (q[1] - p[2]) / 1e6, (p[1] - d[i - 1][2]) / 1e6 if i else 0.0))
print(open(out + '/per_step.txt').read())
PY
