import os, sys, torch, time
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.utils.preproc import qwen_preprocess_video, resize_frames_u8, smart_resize
def timeit(fn, n=20):
    for _ in range(3): fn()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t0=time.perf_counter()
    st.record()
    for _ in range(n): fn()
    en.record(); en.synchronize()
    return st.elapsed_time(en)/n, (time.perf_counter()-t0)/n*1e3
for rep in range(2):
  for (T,H,W,mp) in [(16,480,854,384*784),(16,720,1280,384*784),(16,448,448,384*784)]:
    f = torch.randint(0,256,(T,H,W,3),dtype=torch.uint8,device="cuda")
    h,w = smart_resize(H,W,28,4*784,mp)
    a = timeit(lambda: resize_frames_u8(f,h,w))
    r = resize_frames_u8(f,h,w)
    b = timeit(lambda: qwen_preprocess_video(r, max_pixels=mp))
    c = timeit(lambda: qwen_preprocess_video(f, max_pixels=mp))
    print((T,H,W),(h,w),"resize %.3f/%.3f patchify %.3f/%.3f all %.3f/%.3f"%(a+b+c))
