import csv,glob,os,sys
import numpy as np
f=sorted(glob.glob("/root/repo/gpurun_out/ks_c2/runc/*kernel_trace.csv"), key=os.path.getmtime)[-1]
full=[]
for r in csv.DictReader(open(f)):
    if "k_neighbours2<false, 0>" in r["Kernel_Name"]: full.append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
full=np.array(full); mode=(full>100000).astype(int)
print("steps",len(mode),"fraction single",mode.mean())
s="".join(map(str,mode))
for i in range(0,len(s),150): print(i, s[i:i+150])
