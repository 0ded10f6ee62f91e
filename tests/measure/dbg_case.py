import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import alfred_margaret_amd as am
from oracle import oracle
from tests.helpers import expand_records, oracle_triples
ns=['bAB', 'aB', 'A222aB', 'bABBaa1', 'abaBbAB', 'Baa1abbAB', 'bAB', 'BAABaBAABabAB', '']
hays=['Baa1aBaBababA222A222A222aBBAABa', 'BAABaA222BAABaA222BAABaabab', 'BAABaaBaBab', 'aBBaa1BAABaaBabbABBAABaBaa1', 'BAABaA222Baa1bABBAABaBaa1aBBAABaaB', 'A222Baa1bABA222aBbABBaa1BAABa', 'BAABaBAABaab', 'Baa1Baa1abbABaBBAABaaB']
for case in (0,1):
    o=oracle.Machine(ns); a=am.Automaton(ns)
    exp=oracle_triples(o,case,hays)
    for k in (0,1,2):
        a.set_kernel(k)
        recs=a.run_records(case,hays)
        got=expand_records(o.values_off(),o.values(),recs["haystack"],recs["state"],recs["end_pos"])
        print(case,k,len(recs),got==exp)
        if got!=exp:
            se=set(exp); sg=set(got)
            print(" missing", sorted(se-sg)[:12]); print(" extra", sorted(sg-se)[:12])
            print(" order ok:", sorted(got)==got, len(got), len(exp))
