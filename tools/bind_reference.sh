#!/bin/bash
# bind_reference.sh -- apply INTEGRATION.md's Option A or Option B to a checkout of pelekoudasq/radixHashJoin
# and build its `join` binary against librhj_hip.so.
#
#   tools/bind_reference.sh A|B <reference checkout> <build dir>
#
# The checkout is only read; the edited copies (the edits a maintainer would make in their tree) and
# the binary go to <build dir>.  Nothing of the reference is kept in this repository.
set -euo pipefail
opt=$1; ref=$(realpath "$2"); out=$3
here=$(cd "$(dirname "$0")/.." && pwd)
host=$here/radixhashjoin_amd/host
mkdir -p "$out"; out=$(realpath "$out")
cp "$ref"/*.cpp "$ref"/*.h "$out"/
cd "$out"
CXX=${CXX:-g++}
# the reference's own flags (Makefile:2-3) + the <cstdlib> include its Result.cpp needs on g++ >= 11
FLAGS="-O3 -std=c++11 -pthread -include cstdlib"
case "$opt" in
A)  # Option A: swap three translation units for the mirror (radixhashjoin_amd/host/rhj_compat.{h,cpp})
    rm -f Result.cpp JobScheduler.cpp
    sed -i '86,212d' structs.cpp                      # the hot-path half: histograms, partition, hash_relation, dtors
    printf '#include "rhj_compat.h"\n' > Result.h
    printf '#include "rhj_compat.h"\n' > JobScheduler.h
    # structs.h: tuple / relation / relation_info now come from the mirror
    sed -i '/^struct tuple {/,/^#endif/{/^#endif/!d}' structs.h
    sed -i 's/^#endif/#include "rhj_compat.h"\n#endif/' structs.h
    $CXX $FLAGS -I"$host" -I"$here/include" *.cpp "$host/librhj_compat.a" \
        -L"$here/radixhashjoin_amd" -lrhj_hip -Wl,-rpath,"$here/radixhashjoin_amd" -Wl,-rpath,'$ORIGIN/../../radixhashjoin_amd' -o join
    ;;
B)  # Option B: every reference file kept; only the body of the seam (Result.cpp:90-124) is replaced ...
    { sed -n '1,89p' "$ref/Result.cpp"; cat "$host/binding_option_b.inc"; sed -n '125,$p' "$ref/Result.cpp"; } > Result.cpp
    # ... and the two-line fix of JobScheduler::stop that INTEGRATION.md gives with Option B: `done` is published under
    # queueLock (as shipped -- JobScheduler.cpp:140-146 -- an idle worker can miss the only wake-up, and behind the GPU seam
    # the inner workers are idle from init to stop)
    grep -q '^void JobScheduler::stop() {$' JobScheduler.cpp && grep -q '^    done = true;$' JobScheduler.cpp
    sed -i '/^void JobScheduler::stop() {$/,/^}$/{s/^    done = true;$/    pthread_mutex_lock(\&queueLock);\n    done = true;/; s/^    pthread_cond_broadcast(&cond_nonempty);$/    pthread_cond_broadcast(\&cond_nonempty);\n    pthread_mutex_unlock(\&queueLock);/}' JobScheduler.cpp
    [ "$(grep -c 'pthread_mutex_unlock(&queueLock);$' JobScheduler.cpp)" = 1 ]       # exactly the one line added, inside stop()
    $CXX $FLAGS -I"$here/include" *.cpp -L"$here/radixhashjoin_amd" -lrhj_hip \
        -Wl,-rpath,"$here/radixhashjoin_amd" -Wl,-rpath,'$ORIGIN/../../radixhashjoin_amd' -o join
    ;;
*)  echo "usage: $0 A|B <reference checkout> <build dir>" >&2; exit 2;;
esac
echo "built $out/join (Option $opt)"
