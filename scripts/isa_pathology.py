import sys,re,collections,subprocess
sys.argv=['x',sys.argv[1],sys.argv[2]]
__file__='/root/repo/scripts/isa_by_source.py'
exec(open('/root/repo/scripts/isa_by_source.py').read().split("if span:")[0])
c=collections.defaultdict(collections.Counter)
for ln,key,ins in rows:
    op=ins.split()[0]
    c[key]['n']+=1
    if op in('s_and_saveexec_b64','s_andn2_saveexec_b64','s_or_b64','s_andn2_b64','s_xor_b64','s_and_b64','s_mov_b64','s_cbranch_execz','s_cbranch_execnz','s_branch','s_or_saveexec_b64'): c[key]['ctl']+=1
    if op=='s_waitcnt': c[key]['wait']+=1
    if op.startswith('ds_read'): c[key]['ldsr']+=1
    if op.startswith('ds_write'): c[key]['ldsw']+=1
    if op in('v_readlane_b32','v_writelane_b32','v_readfirstlane_b32'): c[key]['lane']+=1
    if op=='v_mov_b32_e32': c[key]['mov']+=1
    if op.startswith(('flat_','global_','scratch_','buffer_')): c[key]['vmem']+=1
print('%-26s %6s %5s %5s %5s %5s %5s %5s %5s'%('function','instr','ctl','wait','ldsr','ldsw','lane','mov','vmem'))
for k,v in sorted(c.items(),key=lambda kv:-kv[1]['n'])[:45]:
    print('%-26s %6d %5d %5d %5d %5d %5d %5d %5d'%(k[:26],v['n'],v['ctl'],v['wait'],v['ldsr'],v['ldsw'],v['lane'],v['mov'],v['vmem']))
