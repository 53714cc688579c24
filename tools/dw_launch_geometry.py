"""Launch geometry of the depthwise kernels on the supernet's shapes under the ROUND-1 worker rule (workers = total slots / slabs with
the XCD-aware decode): workgroups that land on the fullest XCD against its resident slots.  rounds > 1.00 = a partial second round,
i.e. up to 2x the launch time with persistent workers.  Evidence for DESIGN.md section 5.1 item 1 (-> profiles/r02_dw_launch_geometry.txt).

    python tools/resusage.py atomnas_amd/csrc/dwconv.hip | grep DF16b | awk '{print $1, $3, $5, $7, $9}' > /tmp/occ.txt
    python tools/dw_launch_geometry.py
"""
def fdiv(a,b): return a//b
def cdiv(a,b): return -((-a)//b)
def pad8(c): return (c+7)//8*8
def slab_width(pref,cpad):
    need = 8 if cpad<=8 else (16 if cpad<=16 else (32 if cpad<=32 else 64))
    return min(pref,need)
def lds_pitch(lw,cb):
    rp=lw*cb; return rp+(32-rp%64+64)%64
def statrows(c): return 1024 if c<=64 else (512 if c<1024 else 128)
occ_f={}  # (K,S,CB,TM)->occ
occ_b={}
import re
for l in open('/tmp/occ.txt'):
    m=re.match(r"k_dwconv_(fwd|bwd)IDF16bLi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi0E (\d+) \d+ \d+ (\d+)",l)
    if m:
        d=occ_f if m.group(1)=='fwd' else occ_b
        d[(int(m.group(2)),int(m.group(3)),int(m.group(5)),int(m.group(6)))]=int(m.group(8))
def fwd(N,H,C,K,S):
    P=(K-1)//2; Ho=(H+2*P-K)//S+1; cpad=pad8(C)
    cb_rule=16
    if S==2 and C<=96: cb_rule=8
    elif S==2 and K>=5 and H<=56: cb_rule=8
    elif S==1 and C<=32: cb_rule=32
    elif S==1 and H==28 and K<=5: cb_rule=32
    cb=slab_width(64 if Ho<=7 else cb_rule,cpad)
    TH=min(Ho,14); TW=(min(Ho,14)+6)//7*7
    ty=cdiv(Ho,TH); tx=cdiv(Ho,TW)
    LH=(TH-1)*S+K; LW=(TW-1)*S+K; RP=lds_pitch(LW,cb)
    nslabs=cdiv(cpad,cb); lds=(LH*RP+K*K*cb+8*cb)*4
    tm=7 if (TH<=7 and TW<=7) else 14
    occ=occ_f[(K,S,cb,tm)]
    per_cu=min(occ,160*1024//lds,8)
    return dict(cb=cb,nslabs=nslabs,lds=lds,per_cu=per_cu,ntiles=N*ty*tx,rows=statrows(C))
def bwd(N,H,C,K,S):
    P=(K-1)//2; SW=14 if S==2 else 7; cpad=pad8(C)
    cb_rule=32 if (S==2 or (H<=7 and K<=5)) else 16
    if S==2 and H==56: cb_rule=16
    elif S==1 and H==28 and K<=5: cb_rule=32
    elif S==1 and H==14 and K==5: cb_rule=32
    small=(S==1 and H<=7)
    if small: cb_rule=32 if K==7 else 64
    cb=slab_width(cb_rule,cpad)
    TH=min(H,14); TW=(min(H,14)+SW-1)//SW*SW
    if S==2 and TH%2: TH+=1
    ty=cdiv(H,TH); tx=cdiv(H,TW)
    LH=fdiv(TH-1+P,S)-cdiv(P-(K-1),S)+1; LW=fdiv(TW-1+P,S)-fdiv(-P,S)+1; RP=lds_pitch(LW,cb)
    nslabs=cdiv(cpad,cb); tm=7 if small else 14
    lds=(LH*RP+K*K*cb+cb*(K*K+2)+3*cb)*4+2*tm*tm*cb*2
    key=(K,S,cb,tm)
    occ=occ_b.get(key,2)
    per_cu=min(occ,160*1024//lds,8)
    return dict(cb=cb,nslabs=nslabs,lds=lds,per_cu=per_cu,ntiles=N*ty*tx,rows=statrows(C))
shapes=[(112,32,3,1),(112,96,3,2),(112,96,5,2),(112,96,7,2),(56,144,3,1),(56,144,5,1),(56,144,7,1),(56,144,3,2),(56,144,5,2),(56,144,7,2),
(28,240,3,1),(28,240,5,1),(28,240,7,1),(28,240,3,2),(28,240,5,2),(28,240,7,2),(14,480,3,1),(14,480,5,1),(14,480,7,1),(14,576,3,1),(14,576,5,1),(14,576,7,1),
(14,576,3,2),(14,576,5,2),(14,576,7,2),(7,1152,3,1),(7,1152,5,1),(7,1152,7,1)]
for name,f in (("fwd",fwd),("bwd",bwd)):
    for H,C,K,S in shapes:
        g=f(256,H,C,K,S)
        want=(256*g['per_cu'])//g['nslabs']; want=min(want,g['rows'],g['ntiles']); want=max(want,1)
        per_xcd=cdiv(want,8)*g['nslabs']; slots=32*g['per_cu']
        tiles_per=g['ntiles']/want
        print("%s H%-3d C%-4d k%d s%d cb%-2d slabs%-3d lds%6d per_cu %d workers %3d  xcd0 WGs %3d / slots %3d -> rounds %.2f  tiles/worker %.1f" % (name,H,C,K,S,g['cb'],g['nslabs'],g['lds'],g['per_cu'],want,per_xcd,slots,per_xcd/slots,tiles_per))
