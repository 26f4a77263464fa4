cd keras-segmentation-deeplab-v3.1_amd/csrc
cp pwgemm.hip /tmp/pw_dbg.hip
echo "== nt stores (baseline)"; (cd ../..; timeout 100 python tools/_gemm_dbg.py 2>&1 | grep "M131072")
python - <<'PY'
s=open('pwgemm.hip').read()
old="            __builtin_nontemporal_store(v, &P.c[(size_t)row * P.ldc + col]);\n            st1[j] += v;"
assert s.count(old)==1
open('pwgemm.hip','w').write(s.replace(old,"            P.c[(size_t)row * P.ldc + col] = v;\n            st1[j] += v;"))
PY
make 2>&1 | grep -i error; echo "== plain stores"; (cd ../..; timeout 100 python tools/_gemm_dbg.py 2>&1 | grep "M131072")
cp /tmp/pw_dbg.hip pwgemm.hip
python - <<'PY'
s=open('pwgemm.hip').read()
old="            __builtin_nontemporal_store(v, &P.c[(size_t)row * P.ldc + col]);\n            st1[j] += v;"
assert s.count(old)==1
open('pwgemm.hip','w').write(s.replace(old,"            if (v == 1.2345e-30f) P.c[(size_t)row * P.ldc + col] = v;\n            st1[j] += v;"))
PY
make 2>&1 | grep -i error; echo "== no stores"; (cd ../..; timeout 100 python tools/_gemm_dbg.py 2>&1 | grep "M131072")
