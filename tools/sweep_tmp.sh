for i in 1 2; do tools/ab_run.sh 12; done
