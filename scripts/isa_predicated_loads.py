import sys,re,collections
__file__='/root/repo/scripts/isa_by_source.py'
so,stage=sys.argv[1],sys.argv[2]
sys.argv=['x',so,stage]
exec(open(__file__).read().split("if span:")[0])
# rows: (line, key, instr)
c=collections.Counter(); ex=collections.defaultdict(list)
n=len(rows)
for i in range(n-2):
    a,b=rows[i][2].split()[0],rows[i+1][2].split()[0]
    if a in('s_and_saveexec_b64','s_mov_b64') and 'exec' in rows[i][2] or a=='s_and_saveexec_b64':
        # look ahead up to 3 instrs for a load then exec restore
        for j in range(i+1,min(i+5,n)):
            op=rows[j][2].split()[0]
            if op.startswith(('ds_read','global_load','flat_load')):
                for k in range(j+1,min(j+4,n)):
                    if rows[k][2].startswith('s_or_b64 exec'):
                        c[(rows[j][1],rows[j][0])]+=1
                        break
                break
            if op.startswith(('s_cbranch','v_','s_or')) and not op.startswith('v_mov'): break
tot=collections.Counter()
for (f,l),v in c.items(): tot[f]+=v
print(stage,'predicated single loads by function:',tot.most_common(25))
for (f,l),v in sorted(c.items(),key=lambda kv:-kv[1])[:25]: print('  %-24s line %5d  x%d'%(f,l,v))
