for e in 1 0 1 0; do
  echo "== HOIG_HALO_NMAJOR=$e"
  HOIG_HALO_NMAJOR=$e timeout 120 python tools/time_halo.py 2>&1 | grep "B="
done
