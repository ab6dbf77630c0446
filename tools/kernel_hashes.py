"""md5 of the disassembly of every gfx950 kernel embedded in the built library -> JSON.  Two runs around a host-only edit prove that
no device code moved (the round-4 clean-ups after the last GPU visit were checked this way): python tools/kernel_hashes.py out.json"""
import struct,hashlib,sys,subprocess,tempfile,re
blob=open(__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))), 'yolov6_amd', 'lib', 'libyolov6_hip.so'),'rb').read()
pos=blob.find(b"\x7fELF",1); out={}
while pos>=0:
    e_shoff,=struct.unpack_from("<Q",blob,pos+0x28); es,en=struct.unpack_from("<HH",blob,pos+0x3A); em,=struct.unpack_from("<H",blob,pos+0x12)
    size=e_shoff+es*en
    if em==224 and size>0:
        with tempfile.NamedTemporaryFile(suffix='.co') as f:
            f.write(blob[pos:pos+size]); f.flush()
            txt=subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objdump','-d','--mcpu=gfx950','--no-show-raw-insn',f.name],capture_output=True,text=True).stdout
        cur=None; buf=[]
        for l in txt.split('\n'):
            m=re.match(r'^[0-9a-f]+ <(.*)>:',l)
            if m:
                if cur: out[cur]=hashlib.md5('\n'.join(buf).encode()).hexdigest()
                cur=m.group(1); buf=[]
            elif cur:
                buf.append(re.sub(r'//.*','',l).strip())
        if cur: out[cur]=hashlib.md5('\n'.join(buf).encode()).hexdigest()
    pos=blob.find(b"\x7fELF",pos+4)
import json; json.dump(out,open(sys.argv[1],'w'),indent=0); print(len(out),"kernels")
