export GPU_MAX_HW_QUEUES=8
python3 tools/rank_balance.py field_4k 4 45,90,135
python3 tools/rank_balance.py field_4k 2 90,180,270
python3 tools/rank_balance.py field_4k 8 36,45,68
