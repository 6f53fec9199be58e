"""Developer check on the ping-pong GEMM kernels' ISA: hipcc must not add a vmcnt wait of its own inside the K loop (one compiler-inserted
s_waitcnt vmcnt(0) there drains the LDS-DMA pipeline every iteration: +3.5 ms per step when an epilogue change left loads "pending" on a
divergent path), and spill (scratch) instructions are counted. Usage:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -mllvm -amdgpu-mfma-vgpr-form=1 -S --cuda-device-only -o /tmp/g2.s pianobart_amd/csrc/pb_gemm2.hip
  python tools/check_gemm_waits.py /tmp/g2.s"""
import re,sys
lines=open(sys.argv[1]).read().split('\n')
for kn in ['ILb1ELb0ELi4','ILb1ELb1ELi4','ILb0ELb0ELi4','ILb0ELb1ELi4','ILb1ELb0ELi3','ILb1ELb1ELi3','ILb0ELb0ELi3','ILb0ELb1ELi3']:
    i0=next(i for i,l in enumerate(lines) if re.match(r'^_ZN.*gemm3_kernel'+kn+'.*:',l))
    i1=next(i for i in range(i0,len(lines)) if 's_endpgm' in lines[i])
    mf=[i for i in range(i0,i1) if 'v_mfma' in lines[i]]
    bad=[(i-i0,lines[i].strip()) for i in range(mf[0]-60,mf[-1]+5) if 's_waitcnt' in lines[i] and 'vmcnt' in lines[i] and 'ASMSTART' not in lines[i-1]]
    sp=sum(1 for i in range(i0,i1) if 'scratch_' in lines[i])
    print(kn,'compiler-inserted vmcnt waits near the K loop:',bad,'scratch ops',sp)
