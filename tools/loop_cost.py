#!/usr/bin/env python3
"""Static VALU issue-cycle estimate of the largest loop of a kernel in a hipcc -S listing:
    python tools/loop_cost.py listing.s [mangled kernel name]
4 clocks per wave64 VALU instruction, 16 for quarter-rate ones (transcendentals, 32x32 multiplies)."""
import sys,re
name=sys.argv[2] if len(sys.argv)>2 else '_ZN2te15substeps_kernelILi0ELb1ELb1EEEvNS_6ParamsEPKfNS_7FillJobE'
txt=open(sys.argv[1]).read()
i=txt.index(name+":"); j=txt.index("s_endpgm",i)
lines=txt[i:j].splitlines()
# find first loop header and its back-edge
best=None
for hdr in [k for k,l in enumerate(lines) if 'Loop Header' in l]:
    lab=lines[hdr].split(':')[0]
    ends=[k for k,l in enumerate(lines) if re.search(r's_c?branch\w*\s+'+re.escape(lab)+r'\s*$',l)]
    if ends and (best is None or ends[-1]-hdr>best[1]-best[0]): best=(hdr,ends[-1])
hdr,end=best
quarter={'v_mul_lo_u32','v_mul_hi_u32','v_mad_u64_u32','v_rcp_f32_e32','v_rsq_f32_e32','v_sqrt_f32_e32','v_sin_f32_e32','v_cos_f32_e32','v_log_f32_e32','v_exp_f32_e32','v_rcp_f32_e64','v_rsq_f32_e64'}
n=0;cyc=0;movs=0;pk=0
for l in lines[hdr:end+1]:
    t=l.split()
    if not t or not t[0].startswith('v_'): continue
    n+=1; cyc+=16 if t[0] in quarter else 4
    if t[0].startswith('v_mov') or t[0].startswith('v_pk_mov'): movs+=1
    if t[0].startswith('v_pk_'): pk+=1
print(f"loop VALU instrs {n}, est cycles {cyc}, movs {movs}, packed {pk}")
