cd /root/repo
for t in "fuzz_batch.py 300 241" "fuzz_parity.py 1000 242" "fuzz_large_n.py 243 150" "fuzz_api_order.py 244 3000" "fuzz_strips_policy.py 245 40" "fuzz_rows.py 600 246" "fuzz_batch_rows.py 247 120" "fuzz_pyramid.py 248 120"; do
  echo "=== $t"; ( time timeout 1500 python tools/$t 2>&1 | grep -E "disagree|DISAGREE|Error|error|Traceback" | tail -6 ) 2>&1 | grep -v "^$" | tail -8
done
