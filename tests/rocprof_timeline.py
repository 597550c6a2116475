import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'rgb_to_lab' in r['Kernel_Name']]
s=idx[-1]
t0=int(rows[s]['Start_Timestamp'])
for r in rows[s:]:
    st=int(r['Start_Timestamp'])-t0; en=int(r['End_Timestamp'])-t0
    print("%9.1f %9.1f %8.1f  %s grid=%s wg=%s lds=%s vgpr=%s" % (st/1e3,en/1e3,(en-st)/1e3,r['Kernel_Name'][:58],str(int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']))+'x'+r['Grid_Size_Y'],r['Workgroup_Size_X'],r.get('LDS_Block_Size'),r.get('VGPR_Count')))
