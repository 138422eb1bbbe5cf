set -e
wl=cornell_1920x1080_1024spp_depth8_mis
bash tools/ab_opt.sh slice_iters "512 1024 2048 256 0" $wl 512 1
bash tools/ab_opt.sh lpt_prio "2 0 1" $wl 512 1
bash tools/ab_opt.sh sched_mask "31 63 127 15" $wl 512 1
