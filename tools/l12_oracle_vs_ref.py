import ctypes, numpy as np, subprocess, os, sys, struct
ROOT='/root/repo'
lib=ctypes.CDLL(ROOT+'/oracle/_build/liboracle.so')
L12_DT=np.dtype([("sb","<f8",(2,3,12,32)),("ltmin","<f8",(2,32)),("scalar","<i4",(2,3,32)),("j_scale","<i4",(3,32)),("scfsi","<i4",(2,32)),("bit_alloc","<i4",(2,32)),("mode","<i4"),("mode_ext","<i4"),("jsbound","<i4"),("sblimit","<i4"),("adb_left","<i4"),("crc","<i4"),("magic","<i4"),("frame_index","<i4")])
lib.mp3o_encode_pcm_l12.restype=ctypes.c_size_t
lib.mp3o_encode_pcm_l12.argtypes=[ctypes.c_int]*4+[ctypes.c_char_p,ctypes.c_void_p,ctypes.c_size_t,ctypes.POINTER(ctypes.c_void_p),ctypes.c_void_p,ctypes.c_int]
libc=ctypes.CDLL('libc.so.6'); libc.free.argtypes=[ctypes.c_void_p]
def oracle(layer,rate,kbps,ch,mode,pcm,nd):
    out=ctypes.c_void_p(); d=np.zeros(nd,L12_DT)
    n=lib.mp3o_encode_pcm_l12(layer,rate,kbps,ch,mode.encode(),pcm.ctypes.data,pcm.size,ctypes.byref(out),d.ctypes.data,nd)
    assert out.value
    b=ctypes.string_at(out.value,n); libc.free(out); return b,d
def ref(layer,rate,kbps,mode,pcm):
    with open('in.wav','wb') as f:
        f.write(b'\0'*44); f.write(pcm.astype('<i2').tobytes())
    r=subprocess.run([ROOT+'/oracle/_ref/ref_harness_l12','in.wav','out.mpg',str(layer),str(rate),str(kbps),mode,'dump.bin'],capture_output=True)
    assert r.returncode==0,(r.returncode,r.stderr[-300:])
    return open('out.mpg','rb').read(), np.fromfile('dump.bin',L12_DT)
def sig(n,ch,seed):
    rng=np.random.default_rng(seed)
    t=np.arange(n)
    x=np.zeros((n,ch))
    for c in range(ch):
        f0=rng.uniform(100,4000); 
        x[:,c]=rng.uniform(2000,12000)*np.sin(2*np.pi*f0*t/44100*(1+t/n))+rng.uniform(100,3000)*rng.standard_normal(n)
        x[n//3:n//3+200,c]+=rng.uniform(0,15000)*rng.standard_normal(200)
    return np.clip(x,-32768,32767).astype(np.int16).reshape(-1)
bad=0
cases=[]
for layer in (2,1):
  for rate in (44100,48000,32000):
    for mode in ('s','m','j','d','se','je'):
      ch=1 if mode[0]=='m' else 2
      brs={2:[32,48,56,64,80,96,112,128,160,192,224,256,320,384],1:[32,64,96,128,160,192,224,256,288,320,352,384,416,448]}[layer]
      for kbps in brs:
        cases.append((layer,rate,mode,ch,kbps))
import random
random.seed(1); 
sel=cases if len(sys.argv)>1 and sys.argv[1]=='all' else random.sample(cases,40)
for (layer,rate,mode,ch,kbps) in sel:
    spf=384 if layer==1 else 1152
    n=spf*12+137
    pcm=sig(n,ch,hash((layer,rate,mode,kbps))&0xffff)
    rb,rd=ref(layer,rate,kbps,mode,pcm)
    ob,od=oracle(layer,rate,kbps,ch,mode,pcm,len(rd))
    ok=rb==ob and all(np.array_equal(rd[f],od[f]) for f in L12_DT.names)
    why=''
    if not ok:
        print(len(rd),len(od))
        for f in L12_DT.names:
            if not np.array_equal(rd[f],od[f]):
                fr=[i for i in range(len(rd)) if not np.array_equal(rd[f][i],od[f][i])]
                why+=' %s@%s'%(f,fr[:3])
        bad+=1
    print(layer,rate,mode,kbps,len(rb),len(ob),'OK' if ok else 'DIFF'+why)
print('bad',bad)
