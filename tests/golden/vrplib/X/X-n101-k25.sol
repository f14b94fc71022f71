Route #1: 31 46 35
Route #2: 15 22 41 20
Route #3: 1 70 54
Route #4: 92 9 86
Route #5: 68 90 84 66
Route #6: 76 55 16 69
Route #7: 4 13 74
Route #8: 58 12 5
Route #9: 18 10 39
Route #10: 25 65 78 42 28
Route #11: 7 2 45 43 29 36 72 57
Route #12: 87 37 6 49 14
Route #13: 3 77 63
Route #14: 44 67 88 40
Route #15: 82 60 59
Route #16: 8 17
Route #17: 34 64 96 48 26 47 38
Route #18: 80 94 56 21
Route #19: 71 62 99 98 89
Route #20: 100 61 23
Route #21: 19 97 27
Route #22: 81 51 83
Route #23: 50 91 52
Route #24: 30 85 11 79
Route #25: 75 93
Route #26: 24 95 73 53 33 32
Cost 27591
