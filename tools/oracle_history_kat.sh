#!/bin/bash
# Runs the known-answer tests (tests/test_oracle_settle.py k1..k7 + tests/test_oracle_physics_kat.py) against HISTORICAL versions
# of oracle/settle_ref.c: before and after each edit of the oracle that was motivated by GPU speed.
# The working tree's oracle/settle_ref.c is restored afterwards.
set -e
cd "$(dirname "$0")/.."
cp oracle/settle_ref.c /tmp/settle_ref.current.c
trap 'cp /tmp/settle_ref.current.c oracle/settle_ref.c; python -c "import oracle; oracle.build(True)" >/dev/null 2>&1' EXIT
for rev in 91c1c2e^ 91c1c2e 290ddf3^ 290ddf3 d531545^ d531545 574f050 fa62546 HEAD; do   # 574f050 = end of round 2, fa62546 = persistent manifolds
    if [ "$rev" = HEAD ]; then cp /tmp/settle_ref.current.c oracle/settle_ref.c; else git show "$rev:oracle/settle_ref.c" > oracle/settle_ref.c; fi
    # the trace hook of the current tests is not in the old files: harmless (unused by these tests)
    if python -c "import oracle; oracle.build(True)" >/dev/null 2>&1; then
        r=$(python -m pytest tests/test_oracle_settle.py tests/test_oracle_physics_kat.py -q -p no:cacheprovider 2>&1 | tail -1); f=$(python -m pytest tests/test_oracle_settle.py tests/test_oracle_physics_kat.py -q -p no:cacheprovider 2>&1 | grep "^FAILED" | sed "s/.*:://" | tr "\n" " "); r="$r $f"
    else
        r="does not build against the current headers"
    fi
    echo "$(git log -1 --format=%h "$rev" 2>/dev/null || echo work) ($rev): $r"
done
