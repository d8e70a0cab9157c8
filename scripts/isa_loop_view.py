"""Compact view of a kernel's hottest block (the run of MFMAs) from hipcc -save-temps assembly: one letter per instruction
(M mfma, r/w LDS read/write, G global/buffer load, S store, v/s other vector/scalar, [..] waitcnt, BAR barrier), so a stall shows up
as a wait right behind a load.  usage: isa_loop_view.py <file.s> <kernel-name-substring> [n_mfma_in_loop]"""
import sys, re
lines = open(sys.argv[1]).read().split('\n')
sub = sys.argv[2]
nm = int(sys.argv[3]) if len(sys.argv) > 3 else 48
st = [i for i, l in enumerate(lines) if l.startswith('_Z') and sub in l and ':' in l][0]
en = [i for i in range(st, len(lines)) if 's_endpgm' in lines[i]][0]
body = lines[st:en]
m = [i for i, l in enumerate(body) if 'v_mfma' in l]
end = [i for i, l in enumerate(body) if 's_barrier' in l and i > m[nm - 1]][0]
out = []
for l in body[max(0, m[0] - 140):end + 2]:
    t = l.strip()
    if not t or t.startswith(';'): continue
    op = t.split()[0]
    if op.startswith('v_mfma'): out.append('M')
    elif op.startswith('ds_read'): out.append('r')
    elif op.startswith('ds_write'): out.append('w')
    elif op.startswith(('global_load', 'buffer_load')): out.append('G')
    elif op.startswith(('global_store', 'buffer_store')): out.append('S')
    elif op.startswith('scratch_'): out.append('SCR')
    elif op == 's_waitcnt': out.append('[' + t.split(None, 1)[1].replace('lgkmcnt', 'l').replace('vmcnt', 'v') + ']')
    elif op == 's_barrier': out.append('BAR')
    elif op.startswith(('s_cbranch', 's_branch')): out.append('br')
    elif t.startswith('.LBB'): out.append('\n' + t.split(':')[0] + ':')
    elif op.startswith('v_'): out.append('v')
    elif op.startswith('s_'): out.append('s')
    else: out.append('?' + op)
print(' '.join(out))
print('scratch instructions in the kernel:', sum('scratch_' in l for l in body))
