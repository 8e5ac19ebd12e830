cd rag_project_icd10_amd/csrc/ab3 && timeout 100 ./icd_selftest --oracle $GRAFT_REPO_ROOT/oracle/libicd_oracle.so --skip-cases --bench --iters 1 2>&1 | grep -E "^wg|mode=auto" | head -12
cd .. && for r in 1 2; do timeout 100 ./icd_selftest --oracle $GRAFT_REPO_ROOT/oracle/libicd_oracle.so --skip-cases --bench --iters 20 2>&1 | grep -E "mode=auto|parity"; done
timeout 600 ./icd_selftest --oracle $GRAFT_REPO_ROOT/oracle/libicd_oracle.so 2>&1 | grep -E "FAIL|passed"
