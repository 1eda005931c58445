# LM iterations / s and the Schur phase of the banded cfg-3 scene against the chunk length of its pair lists (PPSFM_BA_CHUNK_LEN)
for L in 32 16 12 8 4; do echo "== chunk length $L"; PPSFM_BA_CHUNK_LEN=$L python tools/nd_probe.py ${1:-500} ${2:-40} 2>&1 | grep -E "LM it/s|phases" | sed -n "2p;4p" | cut -c1-260; done
