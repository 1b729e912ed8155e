import csv,re,sys,glob
def load(tag):
    f=glob.glob(f'/root/repo/gpurun_out/r02c/{tag}/**/*kernel_trace.csv',recursive=True)[0]
    rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:int(r['Start_Timestamp']))
    starts=[i for i,r in enumerate(rows) if 'gn_build_kernel' in r['Kernel_Name']]
    ends=[i for i,r in enumerate(rows) if 'axpy_rev_kernel' in r['Kernel_Name']]
    e=ends[-1]; s=[i for i in starts if i<e][-1]
    step=rows[s:e+1]
    out=[]
    for r in step:
        m=re.search(r'gemm_f64_kernel<(\d+), (\d+), \d+, \d+, (true|false), (true|false), (true|false)',r['Kernel_Name'])
        if m and m.group(3)=='false' and m.group(4)=='false':
            out.append((int(r['Grid_Size_X'])//256, m.group(1), m.group(5), (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
    return out
a=load(sys.argv[1]); b=load(sys.argv[2])
ta=tb=0
for x,y in zip(a,b):
    print(f'grid {x[0]:5d} BM={x[1]} tri={x[2]:5s}  {x[3]:8.1f}  {y[3]:8.1f}  {100*(y[3]/x[3]-1):+6.1f}%'); ta+=x[3]; tb+=y[3]
print('total',ta,tb)
