#!/usr/bin/env python3
"""Re-encode the reference's own test data as compact fixtures (DATA only, no reference source).

Reads (in the build container only; /root/reference does not exist on the GPU box):
  /root/reference/tests/<fn>/small/<case>/data.yaml   (208 c-kzg-4844 vectors, src/lib.rs:23-28)
  /root/reference/trusted_setup.txt                   (4096 G1 + 65 G2, Lagrange form)
Writes next to this script:
  blobs/blob_<k>.bin      the distinct blob byte strings that occur (only 10 of them)
  vectors.json            per-function case list; blobs referenced by index, small fields as hex,
                          expected output (or null == the reference asserts Err, src/lib.rs:47-50)
  trusted_setup_g1.bin    4096 x 48 B compressed G1 (file order)
  trusted_setup_g2.bin    65 x 96 B compressed G2
Fields that are not valid hex are kept verbatim as strings under "raw"; the harness applies the
reference's parse-failure rule (output must be null).
"""
import glob, json, os, sys, hashlib
import yaml

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
FUNCS = ["blob_to_kzg_commitment", "compute_kzg_proof", "compute_blob_kzg_proof",
         "verify_kzg_proof", "verify_blob_kzg_proof", "verify_blob_kzg_proof_batch"]

blob_index = {}
blob_list = []


def blob_ref(hexstr):
    h = hexstr[2:] if hexstr.startswith("0x") else hexstr
    try:
        b = bytes.fromhex(h)
    except ValueError:
        return {"raw": hexstr}
    key = hashlib.sha256(b).hexdigest()
    if key not in blob_index:
        blob_index[key] = len(blob_list)
        blob_list.append(b)
    return {"blob": blob_index[key]}


def main():
    out = {}
    total = 0
    for fn in FUNCS:
        cases = []
        for path in sorted(glob.glob(f"{REF}/tests/{fn}/*/*/data.yaml")):
            with open(path) as f:
                t = yaml.load(f, Loader=getattr(yaml, "CSafeLoader", yaml.SafeLoader))
            name = os.path.basename(os.path.dirname(path))
            inp = {}
            for k, v in t["input"].items():
                if k == "blob":
                    inp[k] = blob_ref(v)
                elif k == "blobs":
                    inp[k] = [blob_ref(x) for x in v]
                else:
                    inp[k] = v
            cases.append({"name": name, "input": inp, "output": t["output"]})
            total += 1
        out[fn] = cases
    os.makedirs(f"{HERE}/blobs", exist_ok=True)
    for i, b in enumerate(blob_list):
        with open(f"{HERE}/blobs/blob_{i}.bin", "wb") as f:
            f.write(b)
    with open(f"{HERE}/vectors.json", "w") as f:
        json.dump({"source": "pawanjay176/kzg_rust tests/ (c-kzg-4844 vectors)", "n_cases": total,
                   "n_blobs": len(blob_list), "functions": out}, f, indent=0)
    lines = open(f"{REF}/trusted_setup.txt").read().split()
    n1, n2 = int(lines[0]), int(lines[1])
    g1 = b"".join(bytes.fromhex(x) for x in lines[2:2 + n1])
    g2 = b"".join(bytes.fromhex(x) for x in lines[2 + n1:2 + n1 + n2])
    assert len(g1) == n1 * 48 and len(g2) == n2 * 96 and (n1, n2) == (4096, 65)
    open(f"{HERE}/trusted_setup_g1.bin", "wb").write(g1)
    open(f"{HERE}/trusted_setup_g2.bin", "wb").write(g2)
    print(f"{total} cases, {len(blob_list)} distinct blobs, sizes {[len(b) for b in blob_list]}")


if __name__ == "__main__":
    main()
