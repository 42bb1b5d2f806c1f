python tools/dev/flight_probe.py 128x160 2 2>/dev/null | grep -v "^-->\|^$" | cut -c1-150
python tools/dev/flight_probe.py 436x1024 2 2>/dev/null | grep -v "^-->\|^$" | cut -c1-150
