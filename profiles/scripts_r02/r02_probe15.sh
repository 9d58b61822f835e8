cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' | head -c 6000; echo
rocprofv3 -L 2>/dev/null | grep -o "\b\(GRBM\|TCP\|TCC\|SPI\)_[A-Z_0-9]*" | sort -u | tr '\n' ' ' | head -c 3000; echo
