"""One reference leg of the key-frame sequence in a process of its own (tests/adapter_sequence_common.py: reference_yardstick): the library pair is chosen by
LDSO_REF_LIB / LDSO_ADAPTER_LIB in the environment (the -O3 build of the reference's translation units cannot share a process with the pin build: same symbols).
    python tests/ref_sequence_worker.py <config> <K> <multithreading 0|1> <out.pkl>"""
import os
import pickle
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    cfg, K, mt, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    from ldso_amd import synth
    from adapter_sequence_common import run_sequence
    win = synth.make_config(cfg, extra_frames=K)
    r, log = run_sequence(win, K, multithreading=bool(mt))
    with open(out, "wb") as f:
        pickle.dump(log, f)
    r.close()


if __name__ == "__main__":
    main()
