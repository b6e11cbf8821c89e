import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'sgd_step_kernel' in r['Kernel_Name']]
a,b=idx[-2]+1, idx[-1]+1
seq=rows[a:b]
print(len(seq),'kernels in the step; step span us', (int(seq[-1]['End_Timestamp'])-int(seq[0]['Start_Timestamp']))/1e3)
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in seq)/1e3
print('busy us', busy)
c=collections.Counter(); t=collections.Counter()
for r in seq:
    n=r['Kernel_Name']
    if 'yolo' in n: continue
    k=n[:110]
    c[k]+=1; t[k]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
for k,v in c.most_common(): print(v, round(t[k],1), k)
# gaps > 10us
gaps=[]
for i in range(1,len(seq)):
    g=(int(seq[i]['Start_Timestamp'])-int(seq[i-1]['End_Timestamp']))/1e3
    if g>8: gaps.append((round(g,1), seq[i-1]['Kernel_Name'][:50], seq[i]['Kernel_Name'][:50]))
print('gaps >8us:', len(gaps), 'total', sum(g[0] for g in gaps))
for g in gaps[:25]: print(g)
