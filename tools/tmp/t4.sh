bash tools/ab_libs.sh "libnghmm_base.so libnghmm_new.so libnghmm_base.so libnghmm_new.so" "c2r" 60 2>&1
bash tools/ab_libs.sh "libnghmm_base.so libnghmm_new.so" "c3 c3r" 20 2>&1
