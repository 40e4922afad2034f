cd /root/repo
for t in "fuzz_batch.py 300 341" "fuzz_parity.py 1000 342" "fuzz_large_n.py 343 150" "fuzz_api_order.py 344 3000" "fuzz_strips_policy.py 345 40" "fuzz_rows.py 600 346" "fuzz_batch_rows.py 347 120" "fuzz_pyramid.py 348 120"; do
  echo "=== $t"; ( time timeout 1500 python tools/$t 2>&1 | grep -E "disagree|DISAGREE|Error|error|Traceback" | tail -6 ) 2>&1 | grep -v "^$" | tail -8
done
