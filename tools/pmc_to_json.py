"""Turns the two rocprofv3 --pmc passes over tools/pmc_gemm.py (FETCH_SIZE, WRITE_SIZE; separate runs, csv output) into the
per-launch HBM traffic record bench.py reads: python tools/pmc_to_json.py <fetch_dir> <write_dir> M N K out.json
Units and corrections as MI355X_MICROARCH.md prescribes (counters in KB; FETCH_SIZE x2 on gfx950)."""
import csv
import glob
import json
import sys


def per_dispatch(d, counter, kernel_sub):
    rows = {}
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get('Counter_Name') == counter and kernel_sub in r.get('Kernel_Name', ''):
                rows[r['Dispatch_Id']] = rows.get(r['Dispatch_Id'], 0.0) + float(r['Counter_Value'])
    vals = list(rows.values())
    return sum(vals) / max(1, len(vals)), len(vals)


def main():
    fdir, wdir, M, N, K, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    fetch_kb, nf = per_dispatch(fdir, 'FETCH_SIZE', 'gemm3_kernel')
    write_kb, nw = per_dispatch(wdir, 'WRITE_SIZE', 'gemm3_kernel')
    fetch_b, write_b = fetch_kb * 1024 * 2, write_kb * 1024
    rec = {
        "kernel": "gemm3_kernel<bf16,NT> (256x256 ping-pong, persistent) fc1 M=%d N=%d K=%d + bias + GELU + derivative out (the epilogue the step launches it with)" % (M, N, K),
        "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB": write_kb, "fetch_bytes_x2_corrected": fetch_b, "write_bytes": write_b,
        "hbm_bytes_per_launch": fetch_b + write_b,
        "algorithmic_bytes": 2 * (M * K + N * K) + 2 * 2 * M * N + 4 * N,
        "launches_averaged": [nf, nw],
        "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/pmc_gemm.py --M=%d; FETCH_SIZE doubled "
                  "(gfx950 counts 128-B requests at 64 B); units KB; tools/pmc_to_json.py" % M,
    }
    json.dump(rec, open(out, 'w'), indent=1)
    print(json.dumps(rec))


if __name__ == '__main__':
    main()
