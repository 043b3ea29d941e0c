repo=$PWD; out=$PWD/gpurun_out/r04; mkdir -p $out
timeout -k 5 120 tools/bin/cumask_probe > $out/cumask_probe.log 2>&1; echo "[r4_f] cumask rc=$?"; head -12 $out/cumask_probe.log; grep -c "more than one" $out/cumask_probe.log
for i in 1 2; do
python3 tools/e2e_leg.py --steps 120 > $out/f_e2e_pf_$i.json 2>$out/f_e2e_pf_$i.err; echo "[r4_f] e2e prefetch: $(cat $out/f_e2e_pf_$i.json) $(tail -2 $out/f_e2e_pf_$i.err)"
python3 tools/e2e_leg.py --steps 120 --no-prefetch > $out/f_e2e_np_$i.json 2>$out/f_e2e_np_$i.err; echo "[r4_f] e2e no prefetch: $(cat $out/f_e2e_np_$i.json)"
python3 tools/e2e_leg.py --steps 120 --resident > $out/f_res_$i.json 2>$out/f_res_$i.err; echo "[r4_f] resident: $(cat $out/f_res_$i.json)"
done
python3 tools/e2e_leg.py --steps 120 --in-flight 4 > $out/f_e2e_pf4.json 2>$out/f_e2e_pf4.err; echo "[r4_f] e2e prefetch 4 lanes: $(cat $out/f_e2e_pf4.json)"
