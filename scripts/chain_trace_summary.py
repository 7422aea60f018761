"""summarise a rocprofv3 kernel trace of scripts/chain_bench.py: far / near inter and intra kernel totals of the last run"""
import csv, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [r for r in rows if 'chain_in' in r['Kernel_Name']]
intra = [r for r in ks if 'intra' in r['Kernel_Name']]
n = len(intra) // 2
t_split = int(intra[n]['Start_Timestamp'])
run2 = [r for r in ks if int(r['Start_Timestamp']) >= t_split]
t0 = min(int(r['Start_Timestamp']) for r in run2); t1 = max(int(r['End_Timestamp']) for r in run2)
print('span ms %.1f kernels %d' % ((t1 - t0) / 1e6, len(run2)))
main = intra[n]['Stream_Id']
far = [r for r in run2 if 'inter' in r['Kernel_Name'] and r['Stream_Id'] != main]
near = [r for r in run2 if 'inter' in r['Kernel_Name'] and r['Stream_Id'] == main]
intr = [r for r in run2 if 'intra' in r['Kernel_Name']]
dur = lambda rs: np.array([int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs]) / 1e3
for name, rs in (('far', far), ('near', near), ('intra', intr)):
    d = dur(rs)
    if len(d) == 0:
        print('%-6s n=    0' % name)
        continue
    print('%-6s n=%5d total %.1f ms mean %.1f us p10 %.1f p50 %.1f p90 %.1f' % (name, len(rs), d.sum() / 1e3, d.mean(), np.percentile(d, 10), np.percentile(d, 50), np.percentile(d, 90)))
d = dur(far); g = np.array([int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']) for r in far])
for i in (10, 100, 500, 1000, 2000, 3000, 4000, 4800):
    if i < len(far): print('far block %4d WGs %4d dur %.1f us' % (i, g[i], d[i]))
for name, rs in (('aux', far), ('main', intr + near)):
    fs = sorted(rs, key=lambda r: int(r['Start_Timestamp']))
    gaps = np.array([int(fs[i + 1]['Start_Timestamp']) - int(fs[i]['End_Timestamp']) for i in range(len(fs) - 1)]) / 1e3
    print('%s gaps: total %.1f ms mean %.1f us p50 %.1f p90 %.1f' % (name, gaps.sum() / 1e3, gaps.mean(), np.percentile(gaps, 50), np.percentile(gaps, 90)))
