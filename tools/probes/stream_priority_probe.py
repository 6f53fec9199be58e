import torch, ctypes
hip = ctypes.CDLL('libamdhip64.so')
lo, hi = ctypes.c_int(), ctypes.c_int()
print('range rc', hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)), 'least', lo.value, 'greatest', hi.value)
for p in (-3, -1, 0, 1, 3):
    try:
        s = torch.cuda.Stream(priority=p); print('torch priority', p, '->', s.priority)
    except Exception as e:
        print('torch priority', p, 'error', repr(e)[:80])
st = ctypes.c_void_p()
print('create low rc', hip.hipStreamCreateWithPriority(ctypes.byref(st), 1, lo.value), st.value)
ext = torch.cuda.ExternalStream(st.value)
print(ext, ext.priority)
