for P in 0 1 3 5; do echo "program $P"; export ETH_KZG_AMD_SLP_PROGRAM=$P; bash tools/sweep_batch.sh 448 576 640 768 1024 1536; done
