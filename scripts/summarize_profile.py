"""Summarise a rocprofv3 --kernel-trace CSV by (kernel, grid): calls, avg/min/max duration.
vgpr = the trace's VGPR_Count (architectural registers per lane), agpr = Accum_VGPR_Count where the trace has
it (gfx950: one unified file of 512 per SIMD lane; a kernel's footprint per lane is their sum)."""
import csv, sys, collections
rows = collections.defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        key = (r['Kernel_Name'], int(r['Grid_Size_X']), int(r['Workgroup_Size_X']), r['VGPR_Count'],
               r.get('Accum_VGPR_Count', ''), r['Scratch_Size'], r['LDS_Block_Size'])
        rows[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
print('kernel,grid_x,wg_x,vgpr,agpr,scratch,lds,calls,avg_us,min_us,max_us')
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    print(f'"{k[0]}",{k[1]},{k[2]},{k[3]},{k[4]},{k[5]},{k[6]},{len(v)},{sum(v)/len(v)/1e3:.1f},{min(v)/1e3:.1f},{max(v)/1e3:.1f}')
