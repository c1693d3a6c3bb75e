import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as entry
pkg = entry.load_package()
try:
    b = pkg.PsdCascadeBank(256)
    print("create with NO torch import: OK")
except Exception as e:
    print("create with NO torch import: FAIL", e)
os.system("cat /proc/%d/maps | grep -i amdhip | awk '{print $6}' | sort -u" % os.getpid())
