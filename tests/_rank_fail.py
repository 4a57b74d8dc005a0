"""a rank that fails: rank 1 exits non-zero, rank 0 would wait forever (launcher test)"""
import os
import sys
import time

if os.environ["RANK"] == "1":
    sys.exit(3)
time.sleep(60)
