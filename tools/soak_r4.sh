cd /root/repo
for t in "fuzz_batch.py 60 41" "fuzz_parity.py 200 42" "fuzz_large_n.py 30 43" "fuzz_api_order.py 600 44" "fuzz_strips_policy.py 12 45" "fuzz_rows.py 120 46" "fuzz_batch_rows.py 20 47" "fuzz_pyramid.py 10 48"; do
  echo "=== $t"; ( time timeout 600 python tools/$t 2>&1 | tail -4 ) 2>&1 | grep -v "^$" | tail -7
done
