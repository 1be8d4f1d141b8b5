for a in 0 1 2 4 8 7 15; do echo "ABL=$a"; COOCC_H2_ABLATE=$a timeout 200 python tools/h2_check.py time 2>&1 | grep -E "con_enc.0|enc.l0 |fpn.out0" | sed 's/.*h2: //' | cut -c1-90; done
