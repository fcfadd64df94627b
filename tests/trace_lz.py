"""Prints the launch-ordered durations of the k_match / k_parse_* kernels of the last run in a rocprofv3 kernel trace."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_match' in r['Kernel_Name'] or 'k_parse' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last run = after the last k_match that follows a k_parse... find last occurrence start of a sequence: k_match then k_parse_spec
idx = [i for i, r in enumerate(rows) if 'k_match' in r['Kernel_Name'] and i + 1 < len(rows) and 'k_parse_spec' in rows[i + 1]['Kernel_Name'] and (i == 0 or 'k_parse_fix' in rows[i - 1]['Kernel_Name'])]
starts = [i for i in idx]
# group runs: a run starts at a k_match preceded by nothing or by a fix AND which is a "first pass" -> take the run containing the last element
line = []
tot = {}
for r in rows[len(rows) // 2:]:
    nm = r['Kernel_Name'].split('(')[0].replace('zada::', '')
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    line.append('%s %.2f' % (nm.replace('k_parse_', 'p_'), d)); tot[nm] = tot.get(nm, 0) + d
print(' | '.join(line))
print({k: round(v, 2) for k, v in tot.items()})
