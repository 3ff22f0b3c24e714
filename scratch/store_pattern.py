import ctypes, os
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "store_pattern.so"))
lib.run.restype = ctypes.c_double
lib.run.argtypes = [ctypes.c_int] * 4
for (M, N) in ((36864, 3072), (36864, 9216), (36864, 12288)):
    for p in (0, 1, 0, 1):
        print(f"M {M} N {N} pattern {p}: {lib.run(p, M, N, 5):.2f} TB/s", flush=True)
