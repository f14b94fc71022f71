Route #1: 99
Route #2: 23 47 46
Route #3: 64 138 102 114
Route #4: 39 89
Route #5: 65 144 131
Route #6: 94 41 78
Route #7: 4 60 51
Route #8: 19 133 141 147
Route #9: 30 56
Route #10: 45 50 9 24 63 11
Route #11: 97 132
Route #12: 124 121 140
Route #13: 38 69 29 25
Route #14: 21 93
Route #15: 43 58 72
Route #16: 143 76 83
Route #17: 87 95
Route #18: 71 5 28
Route #19: 1 59
Route #20: 27 55 22
Route #21: 96 33 36 13 66 145 49
Route #22: 86 122 2
Route #23: 128 146 111
Route #24: 34 139
Route #25: 123 91 79 92
Route #26: 42 134 84
Route #27: 118 107
Route #28: 17 112 70 142
Route #29: 35 113
Route #30: 127 77 109 62
Route #31: 85 110 137
Route #32: 104 32 82
Route #33: 18 75
Route #34: 106 136
Route #35: 88 44
Route #36: 129 10 54 68
Route #37: 14 61 67 98
Route #38: 120 81 7
Route #39: 57 130 26
Route #40: 20 80 119 48 52
Route #41: 105 40
Route #42: 37 125 6 103 101
Route #43: 100 115 135 31
Route #44: 126 53 73
Route #45: 116 90 15
Route #46: 8 3 12 117
Route #47: 16 74 108
Cost 43448
