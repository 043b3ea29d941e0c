import time, os, sys
sys.path.insert(0, os.getcwd())
t0=time.perf_counter()
from font_ocr_amd import Bank
from font_ocr_amd.searcher import Pipeline, Scanner
from font_ocr_amd import _native as N
lib=N.hip()
t1=time.perf_counter(); print("import+load %.1f ms"%((t1-t0)*1e3))
n=lib.focr_device_count(); t2=time.perf_counter(); print("device_count (hip init) %.1f ms"%((t2-t1)*1e3))
s=Scanner(0); t3=time.perf_counter(); print("first ctx %.1f ms"%((t3-t2)*1e3))
s2=Scanner(0); t4=time.perf_counter(); print("second ctx %.1f ms"%((t4-t3)*1e3))
bank=Bank.load("tests/golden/bank_dejavu13_ascii95_x2.bin")
t5=time.perf_counter(); s.set_bank(bank); t6=time.perf_counter(); print("bank upload 1 %.1f ms"%((t6-t5)*1e3))
s2.set_bank(bank); t7=time.perf_counter(); print("bank upload 2 %.1f ms"%((t7-t6)*1e3))
for depth in (1,2):
    ta=time.perf_counter(); p=Pipeline(0,3,depth); tb=time.perf_counter(); p.set_bank(bank); tc=time.perf_counter(); p.close(); td=time.perf_counter()
    print("pipe depth %d: create %.1f ms, bank %.1f ms, close %.1f ms"%(depth,(tb-ta)*1e3,(tc-tb)*1e3,(td-tc)*1e3))
