import csv, collections, sys
rows=list(csv.DictReader(open(sys.argv[1])))
def short(n):
    n=n.split('(')[0]
    for k in ('dist_h','moments','orthobasis','kp_order','grid_scatter','grid_scan','grid_hist','pack_points','rtume','match_prob','match_finalize','hypothesis_gates','index_elementwise','copyBuffer','fillBuffer','arange','direct_copy','CatArray'):
        if k in n: return k
    return n[-30:]
ev=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),short(r['Kernel_Name']),r['Stream_Id'] if 'Stream_Id' in r else '') for r in rows]
ev.sort()
n=len(ev); ev=ev[int(n*0.4):int(n*0.97)]
span=ev[-1][1]-ev[0][0]
busy=0; cur_s,cur_e=ev[0][0],ev[0][1]
gaps=collections.Counter(); gapn=collections.Counter()
prev_name=ev[0][2]
for s,e,nm,_ in ev[1:]:
    if s>cur_e:
        busy+=cur_e-cur_s
        gaps[(prev_name,nm)]+=s-cur_e; gapn[(prev_name,nm)]+=1
        cur_s,cur_e=s,e; prev_name=nm
    else:
        if e>cur_e: cur_e=e; prev_name=nm
busy+=cur_e-cur_s
npairs=max(1,sum(1 for s,e,nm,_ in ev if nm=='dist_h' or 'MatchScratch' in nm or 'coarse_h' in nm))
print('span ms %.3f busy ms %.3f util %.3f pairs %d span/pair %.4f busy/pair %.4f'%(span/1e6,busy/1e6,busy/span,npairs,span/1e6/npairs,busy/1e6/npairs))
for k,v in gaps.most_common(12): print(f"{v/1e3/npairs:8.1f} us/pair  n/pair={gapn[k]/npairs:.2f}  {k[0]} -> {k[1]}")
dur=collections.Counter(); cnt=collections.Counter()
for s,e,nm,_ in ev: dur[nm]+=e-s; cnt[nm]+=1
print()
for k,v in dur.most_common(16): print(f"{v/1e3/npairs:8.1f} us/pair  n/pair={cnt[k]/npairs:.2f} {k}")
# print one pair's sequence
i0=[i for i,x in enumerate(ev) if x[2]=='pack_points'][3]
t0=ev[i0][0]
import os
for s,e,nm,st in ev[i0:i0+int(os.environ.get('GAP_SEQ','34'))]: print(f"  +{(s-t0)/1e3:8.1f} us  dur {(e-s)/1e3:7.1f}  stream {st:>3s}  {nm}")
